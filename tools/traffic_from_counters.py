"""HBM bytes of one training step from the per-kernel counter table of tools/sq_counters.py:
python tools/traffic_from_counters.py profiles/rNN_sq_counters.csv profiles/rNN_traffic.json N_IMAGES
FETCH_SIZE (KB, already x1024 in the table) is doubled ONLY for the kernels whose reads are 16-byte-per-lane streaming loads
(column fetch_x2_applies; MI355X_MICROARCH.md: gfx950 counts those requests at half their bytes); WRITE_SIZE is exact."""
import csv, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[3])
raw = sum(float(r["FETCH_bytes_raw"]) for r in rows)
cor = sum(float(r["FETCH_bytes_raw"]) * (2.0 if r["fetch_x2_applies"] == "1" else 1.0) for r in rows)
wr = sum(float(r["WRITE_bytes"]) for r in rows)
out = {"n_images": n, "kernels_per_step": sum(int(r["calls_per_step"]) for r in rows), "fetch_bytes_raw": raw,
       "fetch_bytes_corrected": cor, "fetch_bytes_upper_x2_all": 2.0 * raw, "write_bytes": wr, "traffic_bytes_per_step": cor + wr,
       "algorithmic_bytes_per_step": 4.841e6 * n, "csrc_sha16": __import__("bench").csrc_sha16(),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py --no-graph (tools/sq_counters.sh); values "
                 "are KB; FETCH_SIZE doubled only for the kernels whose reads are 16-B/lane streaming loads (column fetch_x2_applies of "
                 + sys.argv[1] + "), per MI355X_MICROARCH.md; summed over the kernels of one step"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "method"}))
