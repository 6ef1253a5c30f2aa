python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
