#!/bin/bash
# round 5: retrieval (BASELINE configs[4]) by the collect path against the streaming path, k = 1 / 10 / 50 / 88, and per-kernel times
cd "$(dirname "$0")/../.."
echo "== collect (default)"; python scripts/bench_topk_k.py
echo "== streaming (SLIC_TOPK_COLLECT=0)"; SLIC_TOPK_COLLECT=0 python scripts/bench_topk_k.py
