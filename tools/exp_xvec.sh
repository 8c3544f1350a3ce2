#!/usr/bin/env bash
# Few trees sharing X (the HBM-bound regime): vectorised vs scalar staging of the X tile (gpurun)
for v in 1 0; do echo "DE_X_VEC=$v"; DE_X_VEC=$v python tools/bench_small.py; done
python tools/bench_hbm.py
