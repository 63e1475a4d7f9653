#!/bin/bash
# build (non-dev) and run the CPU suite
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3 && timeout 1500 python -m pytest tests/ -x -q -m "not gpu" 2>&1 | tail -5
