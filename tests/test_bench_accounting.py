"""bench.py's byte accounting against SURVEY.md section 8(d) (no GPU needed): the per-target figures the roofline
fractions are priced with, and the launcher's refusal to run N ranks on fewer than N devices."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_algorithmic_bytes_are_the_surveys_per_target_figures():
    n, e, m, k = 1000, 700, 900, 20
    for c in (1, 3):
        a = bench.algorithmic_bytes(n, e, m, k, c)
        assert a["gather"] == n * (128 + 72 * c)                     # 200 B (C = 1), 344 B (C = 3)
        assert a["knn_query"] == n * 184 + e * 24                    # 24 + 8k, + the centroids once
        assert a["locate"] == n * 568                                # 24 + 8k + 64 + 192 + 128
        assert a["centroid"] == e * (8 * 8 + 8 * 24 + 24)
        assert a["knn_cell"] == a["knn_query"] and a["locate_pass0"] == a["locate"]
    assert 184 + 568 + 200 == 952                                    # end to end, C = 1


def test_actual_bytes_never_exceed_what_the_survey_prices_per_stage():
    n, e, m, k = 10_077_696, 9_938_375, 10_077_696, 20
    alg = bench.algorithmic_bytes(n, e, m, k, 1)
    act = bench.actual_bytes(n, e, m, k, 1, fused_gather=True)
    for stage in ("centroid", "knn_cell", "locate_pass0", "gather"):
        assert 0 < act[stage] <= alg[stage] + (alg["gather"] if stage == "locate_pass0" else 0)
    # the stand-alone gather: ids + weights + output + the field once
    assert act["gather"] == n * 136 + m * 8


def test_launcher_refuses_more_ranks_than_devices():
    # no GPU in the build container: --gpus 2 must fail loudly (exit code 2) before anything is launched
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("MM_BENCH_REHEARSE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "GPU(s)" in r.stderr
