cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05b
python tools/dump_jobs.py 512 > gpurun_out/r05b/jobs512.txt 2>&1 || { tail -5 gpurun_out/r05b/jobs512.txt; exit 1; }
cat gpurun_out/r05b/jobs512.txt
python -m pytest tests/test_gpu_modules.py tests/test_gpu_engine.py -m gpu -x -q -k "rehearsal or bit_identical or dp_launch" > gpurun_out/r05b/tests.log 2>&1; rc=$?; tail -15 gpurun_out/r05b/tests.log; exit $rc
