"""N > 1 launch path on the GPU box: the ranks are started by the entry points themselves
(child processes, torch.distributed.run), run the REAL device-resident link, and meet in one
all-reduce.  A one-GPU box rehearses N = 2 with both ranks on cuda:0 (WF_BENCH_REHEARSAL=1, gloo
for the collective); the RCCL branch itself runs in a world-size-1 `nccl` group."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _run(cmd, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{cmd}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_two_ranks_through_the_real_block_runner_equal_one_rank():
    """Same SweepPlan dealt to 2 ranks (gpu_block_runner -> wf_link_run on each) and to 1 rank:
    the all-reduced counter table is identical — the shards are disjoint, complete and idempotent."""
    probe = str(ROOT / "tests" / "_rank_probe.py")
    code = "import sys; from waveforms_amd.bert import spawn_ranks; sys.exit(spawn_ranks({!r}, {}, ['--gpu', '--nsym', '262144', '--blocks', '3']))"
    one = _run([sys.executable, "-c", code.format(probe, 1)])
    two = _run([sys.executable, "-c", code.format(probe, 2)])
    assert one["world"] == 1 and two["world"] == 2
    assert one["counts"] == two["counts"]
    c = np.array(one["counts"])
    assert (c[:, 2] == 3 * (262144 - 3)).all()          # every block compares ncols - length = nsym - 3 symbols
    assert c[0, 1] > c[1, 1] > c[2, 1] > 0               # BER falls with Eb/N0 (0, 4, 8 dB)


def test_bench_gpus_2_without_torchrun_rehearsal():
    """`python bench.py --gpus 2` — the command shape the driver uses — starts its own ranks."""
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--nsym", "1000000",
                "--no-cpu-baseline", "--steady-steps", "20"], {"WF_BENCH_REHEARSAL": "1"})
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["ber"]["symbols"] == 2 * 3 * (1000000 - 3)
    assert out["value"] > 0 and "roofline" in out
    # the N > 1 line carries the evidence of its own collective: what backend ran, how many ranks it saw, which device
    # each rank drove, every rank's own counts (their sum is the reduced total) and every rank's own steady state
    c = out["collective"]
    assert c["backend"] == "gloo" and c["world_seen"] == 2 and c["all_reduce_us"] > 0
    assert [d["rank"] for d in c["devices"]] == [0, 1] and all(d["pci"] and d["compared"] == 3 * (1000000 - 3) for d in c["devices"])
    assert c["sum_of_rank_counts_equals_reduced"] is True
    assert sum(d["bit_errors"] for d in c["devices"]) == out["ber"]["bit_errors"]
    assert c["distinct_devices"] == 1                         # the rehearsal: both ranks on cuda:0 (a real run shows N)
    per = out["steady_state_per_rank"]
    assert [p["rank"] for p in per] == [0, 1] and all(p["ms_per_step"] > 0 for p in per)


def test_ber_sweep_tool_gpus_2_equals_single_process(tmp_path):
    args = ["--ebn0", "4:6", "--symbols-per-point", "2e6", "--block", "524288"]
    one = _run([sys.executable, "tools/ber_sweep.py", *args])
    two = _run([sys.executable, "tools/ber_sweep.py", "--gpus", "2", *args], {"WF_BENCH_REHEARSAL": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["counts"] == two["counts"]
    assert two["collective_backend"] == "gloo" and two["world_seen"] == 2 and two["init_seconds"] >= 0
    assert one["collective_backend"] is None and one["world_seen"] == 1


def test_rccl_all_reduce_of_the_counter_table_world1():
    """The `nccl` (= RCCL) branches — init_process_group("nccl", device_id=...) of init_ranks and the
    .cuda() all-reduce of all_reduce_counts — executed once, in a world-size-1 group on cuda:0."""
    code = """
import os, json, socket, numpy as np
with socket.socket() as s:                    # a free rendezvous port, picked like spawn_ranks does
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
from waveforms_amd.bert import init_ranks, all_reduce_counts
import torch, torch.distributed as dist
rank, world, d, dev = init_ranks()
assert d is None and dev == "cuda"           # world 1: init_ranks opens no group; open the nccl one by hand
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
t = np.arange(39, dtype=np.int64).reshape(13, 3) * 1000003
out = all_reduce_counts(t)
x = torch.ones(4, device="cuda"); dist.all_reduce(x, op=dist.ReduceOp.MAX)
dist.barrier(); dist.destroy_process_group()
print(json.dumps({"same": bool((out == t).all()), "dtype": str(out.dtype), "max": float(x.max())}))
"""
    out = _run([sys.executable, "-c", code])
    assert out == {"same": True, "dtype": "int64", "max": 1.0}


def _gpu_count():
    try:
        import torch

        return torch.cuda.device_count()        # (does not initialise HIP on this image)
    except Exception:   # noqa: BLE001
        return 0


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: real RCCL between two ranks")
def test_bench_gpus_2_real_nccl_equals_rehearsal():
    """Two ranks on two GPUs with backend nccl (= RCCL over xGMI): the all-reduced counters equal the one-GPU
    rehearsal's (both ranks on cuda:0, gloo), per-rank times are reported, the CPU baseline stays in the line."""
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--nsym", "1000000",
           "--cpu-sample", "65536", "--cpu-loop-sample", "4096"]
    real = _run(cmd)
    reh = _run(cmd + ["--no-cpu-baseline"], {"WF_BENCH_REHEARSAL": "1"})
    assert real["n_gpus"] == 2 and real["ber"] == reh["ber"]
    assert len(real["per_rank_ms_per_step"]) == 2 and real["rank_time_max_over_min"] >= 1.0
    assert real["cpu_baseline"]["value"] > 0


def test_bench_line_reports_binding_roofline_and_steady_state():
    """One-GPU line: the roofline that binds is the one in bound / frac (HBM kept as hbm_frac), the CPM path is
    asked from the library, and the long steady-state figure sits beside the driver-timed value."""
    out = _run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--steady-steps", "200",
                "--overlap-streams", "0"])
    r = out["roofline"]
    assert r["bound"] in ("hbm", "valu_issue") and r["frac"] == r["binding_frac"] and 0 < r["hbm_frac"] <= 1
    if "valu_issue_frac" in r:
        assert r["frac"] == max(r["valu_issue_frac"], r["hbm_frac"])
    assert out["steady_state"]["steps"] == 200 and out["steady_state"]["value"] > 0
    out = _run([sys.executable, "bench.py", "--waveform", "multih", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                "--steady-steps", "0", "--overlap-streams", "0", "--nsym", "1000000"])
    assert out["roofline"]["kernel"].startswith(("mod_chan_bank_kernel<4, 16, 8>", "cpm_viterbi_kernel<4, 2>"))
