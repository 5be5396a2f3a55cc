#!/bin/bash
# usage (GPU box): tools/sweep_caps.sh -> ms/step for workgroup caps of the MFMA weight-gradient kernels and the tail kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
echo "default: $(run)"
for c in 256 512 768 2048; do echo "CGS_WGRAD_BLOCKS=$c: $(CGS_WGRAD_BLOCKS=$c run)"; done
for c in 256 384 640 768; do echo "CGS_TAIL_BWD_BLOCKS=$c: $(CGS_TAIL_BWD_BLOCKS=$c run)"; done
for c in 512 768 1536; do echo "CGS_TAIL_FWD_BLOCKS=$c: $(CGS_TAIL_FWD_BLOCKS=$c run)"; done
