"""bench.py's host-side pieces that need no GPU: the algorithmic-bytes formula of SURVEY.md section 8(d), the CPU share
detection, the workload table (BASELINE.json configs), and the refusal to run without a GPU (no CPU fallback)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_algorithmic_bytes_formula():
    b = _bench()
    st = {"samples": 1000, "segments": 2500, "nodes_tested": 50000, "prims_tested": 24000, "floor_tested": 0}
    total, trace = b.algorithmic_bytes(st, spp=10, triangles=False)
    assert trace == 50000 * 64 + 24000 * 32 + 2500 * 64
    assert total == 1000 * 24 / 10 + 50000 * 64 + 24000 * 32 + 2500 * 192
    st["floor_tested"] = 100
    total_t, _ = b.algorithmic_bytes(st, spp=10, triangles=True)
    assert total_t == 1000 * 24 / 10 + 50000 * 64 + (24000 + 100) * 84 + 2500 * 192


def test_workloads_are_the_baseline_configs():
    b = _bench()
    cfg = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "north_star" in cfg
    assert b.WORKLOADS["shirley_1080p_spp64_d8"] == ("shirley", 1920, 1080, 64, 8)   # configs[1], the headline
    assert b.WORKLOADS["shirley_600x300_spp32_d8"] == ("shirley", 600, 300, 32, 8)    # configs[0], the README command
    assert b.WORKLOADS["cornell_1024_spp256_d16"][1:] == (1024, 1024, 256, 16)
    assert b.WORKLOADS["shirley_4k_spp256_d8"][1:] == (3840, 2160, 256, 8)
    assert b.HBM_PEAK_GBS == 8000.0


def test_effective_cpus_is_sane():
    n = _bench().effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return  # on the GPU box this is covered by running the benchmark itself
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True)
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stderr + r.stdout)


def _json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


@pytest.mark.parametrize("world,workload", [(2, "shirley_600x300_spp32_d8"), (8, "shirley_1080p_spp64_d8")])
def test_bench_starts_its_own_ranks(world, workload):
    """`python bench.py --gpus N` with no launcher (N = 8 on the headline frame is what the driver's scaling run starts
    first): the parent starts N fresh ranks through torch.distributed.run before touching any GPU, relays rank 0's ONE JSON line and the exit code.  The rehearsal flag replaces the renderer (which
    needs a GPU) by synthetic bands, so the launcher, the rendezvous and the band exchange run here on CPU."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--rehearse-launch",
                        "--steps", "2", "--warmup", "1", "--workload", workload],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == world and lines[0]["steps"] == 2 and lines[0]["warmup"] == 1
    line = lines[0]
    assert line["value"] is None and line["rehearsal"]["bands_arrived_in_place"] is True
    # the N > 1 line is self-proving: what RCCL (here: gloo) saw, the timed frame's parity slot, the CPU leg and -- on the headline
    # workload -- the configuration the scaling target is quoted on, timed across the same ranks (all null-valued in a rehearsal)
    col = line["collective"]
    assert col["world"] == world and col["backend"] == "gloo" and len(col["render_ms_per_rank"]) == world
    assert col["rank0_gather_ms"] > 0 and col["bytes_into_rank0_per_step"] == (world - 1) * col["bytes_per_peer_per_step"] > 0
    assert set(line["parity"]) >= {"timed_frame", "rel_linf_vs_cpu_ref", "tolerance"} and line["parity"]["timed_frame"]["pixels"] == 64
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["kind"] == "port"
    if workload == "shirley_1080p_spp64_d8":
        w5 = line["workloads"]["shirley_4k_spp256_d8"]
        assert w5["n_gpus"] == world and w5["steps"] == 3 and w5["warmup"] == 1 and w5["samples_per_step"] == 3840 * 2160 * 256
        assert "parity" in w5 and w5["collective"]["world"] == world and len(w5["collective"]["render_ms_per_rank"]) == world
    else:
        assert "workloads" not in line


def test_self_launched_ranks_refuse_to_run_without_a_gpu():
    """Without the rehearsal flag the ranks are real: on a box with no GPU each one refuses, and the parent hands the
    failure on (non-zero exit, no JSON line)."""
    import torch
    if torch.cuda.is_available():
        return
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stderr + r.stdout)
    assert _json_lines(r.stdout) == []
