#!/bin/bash
tools/sq_counters.sh c4pmc --mode infer --fp16 --batch 2048
python tools/sq_any.py gpurun_out/c4pmc mask_infer tail_ h5conv > gpurun_out/c4pmc_sq.txt 2>&1
