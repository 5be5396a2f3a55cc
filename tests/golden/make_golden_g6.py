#!/usr/bin/env python3
"""G6: the reference's OWN `main.py -process` run end to end here, its PNG outputs captured as data.

Run in the build container only (the reference does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_g6.py

`/root/reference/main.py` is imported as a module (its CLI lives in `main()` behind an `if __name__ == "__main__"` guard) inside a
scratch working directory and `main.main()` is called with `-process` command lines.  What this process supplies around it:
  * the two shims of make_golden.py (`torchvision` stub modules, `np.int = int`) for `nets.py`;
  * EMPTY stub modules for `minerl`, `cv2` and `ffmpeg` (absent from this image; imported at main.py:15,17,22, never touched by the
    `-process` path: main.py:1103-1223 uses numpy, PIL and the nets classes only);
  * a TrueType file at the relative path `Handler.__init__` opens (main.py:70, `./isy_minerl/segm/etc/Ubuntu-R.ttf`): a copy of
    matplotlib's DejaVuSans.ttf -- the font is only used by the video / visualisation code, not by `-process`;
  * the G1 weights saved as `torch.save(state_dict)` under the checkpoint names the reference's Handler computes itself
    (`H.save_paths`, main.py:86-104), and the source frames as PNG files.
Only DATA is written: g6_process.npz = the source frames, every output file's name and pixel array, for three command lines
(default; `-concatenated`; `--binarymaskthreshold 0.52`).  No reference source is copied."""
import json
import os
import shutil
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

import numpy as np
import torch
from PIL import Image

np.int = int
for name in ("torchvision", "torchvision.models", "minerl", "cv2", "ffmpeg"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, REF)

torch.set_num_threads(1)
torch.use_deterministic_algorithms(True)


def run(tmp, argv):
    import main as refmain          # the reference's CLI module
    old = sys.argv
    sys.argv = ["main.py"] + argv
    try:
        refmain.main()
    finally:
        sys.argv = old


def main():
    import matplotlib
    raw = dict(np.load(os.path.join(HERE, "g1_weights_chfak1.npz")))
    pc = {k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("critic/")}
    pm = {k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("masker/")}
    rs = np.random.RandomState(6)
    n = 5
    X = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    X[1] = (X[1] * 0.3).astype(np.uint8)
    X[2, 10:40, 20:60] = 210                     # a flat patch
    names = ["frame_000", "b.second", "third-frame", "x", "zz_last.v2"]      # stems with dots and dashes (the reference strips the last extension)
    tmp = tempfile.mkdtemp(prefix="g6_")
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        os.makedirs("isy_minerl/segm/etc")
        shutil.copy(os.path.join(matplotlib.get_data_path(), "fonts", "ttf", "DejaVuSans.ttf"), "isy_minerl/segm/etc/Ubuntu-R.ttf")
        os.makedirs("src")
        for nm, x in zip(names, X):
            Image.fromarray(x).save(f"src/{nm}.png")
        # checkpoint names: ask the reference's own Handler
        import argparse
        import main as refmain
        probe = sys.argv
        sys.argv = ["main.py", "--model", "m"]
        # (main() would also load and segment: build the Handler from the parsed defaults instead, only to read save_paths)
        parser_args = None
        real_handler = refmain.Handler

        class Probe(real_handler):
            def __init__(self, args):
                super().__init__(args)
                nonlocal parser_args
                parser_args = self.save_paths
                raise SystemExit(0)
        refmain.Handler = Probe
        try:
            refmain.main()
        except SystemExit:
            pass
        refmain.Handler = real_handler
        sys.argv = probe
        os.makedirs("m/saves")
        torch.save(pc, parser_args["critic"])
        torch.save(pm, parser_args["masker"])
        out = {"frames": X, "names": np.array(names), "checkpoint_names": np.array([os.path.relpath(parser_args["critic"]), os.path.relpath(parser_args["masker"])])}
        runs = {"default": ["-process", "--model", "m", "--source-imgs", "src", "--mask-output-imgs", "out_default"],
                "concat": ["-process", "-concatenated", "--model", "m", "--source-imgs", "src", "--mask-output-imgs", "out_concat"],
                "thr052": ["-process", "--model", "m", "--source-imgs", "src", "--mask-output-imgs", "out_thr052", "--binarymaskthreshold", "0.52"]}
        listing = {}
        for tag, argv in runs.items():
            run(tmp, argv)
            files = sorted(os.listdir("out_" + tag))
            listing[tag] = files
            for f in files:
                out[f"{tag}/{f}"] = np.array(Image.open(f"out_{tag}/{f}"))
        out["listing_json"] = np.array(json.dumps(listing))
        np.savez_compressed(os.path.join(HERE, "g6_process.npz"), **out)
        print("wrote g6_process.npz:", {k: len(v) for k, v in listing.items()}, "files; checkpoints", out["checkpoint_names"].tolist())
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
