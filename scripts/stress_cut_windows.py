"""long_run_case seeds beyond the test suite's through three record forms against the oracle (front.hip, window_cut)."""
import sys
sys.path.insert(0, ".")
from tests.test_gpu_random import long_run_case, _run
first, n = int(sys.argv[1]), int(sys.argv[2])
fails = 0
for seed in range(first, first + n):
    w = long_run_case(seed)
    for grouped, form in ((True, "four"), (True, "marked"), (bool(seed & 1), "packed")):
        try:
            _run(w, grouped, form)
        except AssertionError as e:
            fails += 1
            print("FAIL seed", seed, grouped, form, str(e)[:300].replace("\n", " | "), flush=True)
print("fails", fails, "of", n, "seeds from", first)
