#!/bin/bash
# Everything a reviewer would run on a GPU box, in order: build, CPU suite, GPU parity suite, smoke, default bench.
set -e
cd "$(dirname "$0")/.."
python __graft_entry__.py smoke                # build() then smoke() on cuda:0
python -m pytest tests -q -m "not gpu"
python -m pytest tests -q -m gpu
python bench.py
