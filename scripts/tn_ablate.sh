#!/bin/bash
# Ablations of the weight-gradient product: TAD_MM16_DBG bit 1 = no accumulation into C, 2 = no MFMAs, 4 = no ring refill.
for d in ${@:-1 2 3 5 6 7}; do echo "== DBG=$d"; TAD_MM16_DBG=$d timeout -k 10 120 python scripts/mm16_bench.py tn 2>&1 | tail -8 || exit 1; done
