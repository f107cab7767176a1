"""What `bench.py --gpus N` runs (no GPU needed): N = 1 is BASELINE configs[2]; N >= 2 is the north star's workload
(strong scaling on 4800x1600 at Re 200) whenever a rank's share fits its HBM, otherwise weak scaling."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_default_workloads_per_gpu_count():
    B = _bench()
    assert B.choose_workload(1) == ("weak", 1200, 400, 100.0)
    assert B.choose_workload(2) == ("weak", 2400, 400, 100.0)          # 327 GB per rank: the strong case does not fit
    assert B.choose_workload(4) == ("strong", 4800, 1600, 200.0)       # 164 GB per rank
    assert B.choose_workload(8) == ("strong", 4800, 1600, 200.0)       # 82 GB per rank
    # a rank's share of the strong-scaling case: below 0.8 x 288 GB and below 2^31 non-zeros in block (0,0)
    for n in (4, 8):
        assert B.BYTES_PER_DOF * B.N_DOFS_NORTH_STAR / n <= 0.8 * B.HBM_BYTES
        assert 428_350_320 * 16 / n < 2 ** 31


def test_explicit_arguments_win():
    B = _bench()
    assert B.choose_workload(8, "weak") == ("weak", 9600, 400, 100.0)
    assert B.choose_workload(2, "strong") == ("strong", 4800, 1600, 200.0)
    assert B.choose_workload(4, "auto", "600,200", 50.0) == ("weak", 2400, 200, 50.0)
    assert B.choose_workload(4, "strong", "600,200") == ("strong", 600, 200, 100.0)
