"""CPU, world_size 2 over gloo: the N>1 path of the engine is a static partition + barrier + max-over-ranks time."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_samples, ens, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from dg_tta_amd.sharding import max_over_ranks, rank_world, units_for_rank
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert rank_world() == (rank, world)
    mine = units_for_rank(num_samples, ens, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    dist.barrier()
    t = max_over_ranks(1.0 + rank)                     # rank 1 is "slower"
    if rank == 0:
        out.put((gathered, t))
    dist.destroy_process_group()


def test_partition_is_disjoint_complete_and_time_is_max():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    flat = [u for part in gathered for u in part]
    assert sorted(flat) == [(s, e) for s in range(5) for e in range(3)]          # complete
    assert len(set(flat)) == len(flat)                                          # disjoint
    assert {s for s, _ in gathered[0]} == {0, 2, 4} and {s for s, _ in gathered[1]} == {1, 3}
    assert t == 2.0


def test_single_process_defaults():
    from dg_tta_amd.sharding import max_over_ranks, units_for_rank
    assert units_for_rank(2, 2, 0, 1) == [(0, 0), (0, 1), (1, 0), (1, 1)]
    assert max_over_ranks(0.5) == 0.5


def test_pairs_are_sharded_when_samples_are_fewer_than_ranks():
    """with fewer samples than GPUs the (sample, ensemble) PAIRS are dealt out; the partition stays
    complete and disjoint, and member 0's owner (who runs the sample's ensemble inference) is well defined."""
    from dg_tta_amd.sharding import unit_owner, units_for_rank
    for ns, ens, world in [(1, 3, 8), (2, 3, 4), (3, 3, 8), (8, 3, 8), (5, 1, 2), (1, 1, 2)]:
        parts = [units_for_rank(ns, ens, r, world) for r in range(world)]
        flat = [u for p in parts for u in p]
        assert sorted(flat) == [(s, e) for s in range(ns) for e in range(ens)] and len(set(flat)) == len(flat)
        if ns >= world:
            assert all(len({unit_owner(s, e, ns, ens, world) for e in range(ens)}) == 1 for s in range(ns))
        else:       # pairs spread: no rank idles while another holds two units, as far as the unit count allows
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1
    assert units_for_rank(1, 3, 0, 8) == [(0, 0)] and units_for_rank(1, 3, 2, 8) == [(0, 2)]


def _barrier_worker(rank, world, port, run_dir, out):
    """Two ranks on one run directory: each 'adapts' its units (writes the parameter file under a temporary name and
    renames it), rank 0 waits for rank 1's files and marker before it lists the directory for the summary."""
    import time
    from pathlib import Path
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from dg_tta_amd.sharding import done_marker, mark_rank_done, units_for_rank, wait_for_files
    dist.init_process_group("gloo", rank=rank, world_size=world)
    run_dir = Path(run_dir)
    units = [(s, e) for s in range(1) for e in range(2)]
    for (s, e) in units_for_rank(1, 2, rank, world):
        if rank == 1:
            time.sleep(1.0)                         # the other GPU is slower
        tmp = run_dir / f"case{s}__ensemble_idx_{e}.pt.tmp{rank}"
        tmp.write_bytes(b"x" * 1000)
        tmp.replace(run_dir / f"case{s}__ensemble_idx_{e}.pt")
    if rank == 0:       # owner of member 0: needs every member of the sample
        wait_for_files([run_dir / f"case{s}__ensemble_idx_{e}.pt" for s, e in units], timeout_s=60)
        (run_dir / "case0.npy").write_bytes(b"prediction")
    mark_rank_done(run_dir, rank)
    if rank == 0:
        wait_for_files([done_marker(run_dir, r) for r in range(world)], timeout_s=60)
        out.put(sorted(p.name for p in run_dir.iterdir() if not p.name.startswith(".")))
    dist.barrier()
    dist.destroy_process_group()


def test_filesystem_barrier_two_ranks(tmp_path):
    import pytest
    from dg_tta_amd.sharding import wait_for_files
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_barrier_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    listing = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert listing == ["case0.npy", "case0__ensemble_idx_0.pt", "case0__ensemble_idx_1.pt"]
    with pytest.raises(TimeoutError, match="never_written"):
        wait_for_files([tmp_path / "never_written.pt"], timeout_s=0.3, poll_s=0.05)


def test_fanout_parent_reports_failed_children(tmp_path):
    """run.py's parent: a child killed by a signal (negative return code) must not read as success, and the surviving
    children are stopped instead of waiting in the filesystem barrier."""
    import subprocess
    import sys
    import time
    from dg_tta_amd.run import _wait_children
    from dg_tta_amd.sharding import child_devices
    ok = subprocess.Popen([sys.executable, "-c", "pass"])
    killed = subprocess.Popen([sys.executable, "-c", "import os, signal; os.kill(os.getpid(), signal.SIGABRT)"])
    waiting = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
    t0 = time.monotonic()
    rcs = _wait_children([ok, killed, waiting], poll_s=0.05)
    assert time.monotonic() - t0 < 60
    assert rcs[0] == 0 and rcs[1] < 0 and rcs[2] != 0
    assert max(rcs) == 0          # what the old `max(p.wait())` would have reported
    old = os.environ.get("HIP_VISIBLE_DEVICES")
    try:
        os.environ["HIP_VISIBLE_DEVICES"] = "4,5,6,7"
        assert child_devices(2) == ["4", "5"]
        os.environ.pop("HIP_VISIBLE_DEVICES")
        assert child_devices(3) == ["0", "1", "2"]
    finally:
        if old is not None:
            os.environ["HIP_VISIBLE_DEVICES"] = old


def test_bench_launches_its_own_ranks(tmp_path):
    """`bench.py --gpus 2` without RANK / WORLD_SIZE starts two ranks itself: the line says n_gpus 2, the
    whole-job value is twice one instance's, the time is the slowest rank's.  Stub runner: no GPU is touched."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--stub-runner", "0.05", "--steps", "3",
                          "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE line, printed by rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["rendezvous"] == "gloo" and out["config"]["world_size_seen"] == 2      # which backend really ran
    assert len(out["per_rank_epochs_per_s"]) == 2
    assert out["value"] == pytest.approx(2 * out["value_per_gpu"], rel=1e-4)
    assert out["value_per_gpu"] <= min(out["per_rank_epochs_per_s"]) * 1.0001      # quoted on the slowest rank
    assert "own child" in out["config"]["launcher"]
    # one rank under an external launcher (RANK / WORLD_SIZE set): no children are started
    env1 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--stub-runner", "0.01", "--steps", "1",
                          "--warmup", "0"], env=env1, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and json.loads(res.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_launcher_reports_a_failed_rank():
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--stub-runner", "-1"], env=env,
                         capture_output=True, text=True, timeout=300)        # a negative sleep raises in every rank
    assert res.returncode != 0 and "rank return codes" in res.stderr


def test_barrier_aborts_on_a_failed_peer_and_stale_markers_are_cleared(tmp_path):
    from dg_tta_amd.sharding import done_marker, failed_marker, mark_rank_done, mark_rank_failed, wait_for_files
    mark_rank_failed(tmp_path, 1)
    with pytest.raises(RuntimeError, match="peer rank failed"):
        wait_for_files([tmp_path / "x.pt"], timeout_s=30, poll_s=0.05, abort_if=[failed_marker(tmp_path, r) for r in range(2)])
    mark_rank_done(tmp_path, 0)
    assert done_marker(tmp_path, 0).is_file()


def test_failed_marker_of_an_earlier_launch_does_not_abort_a_resumed_run(tmp_path, monkeypatch):
    """a `.rank_k.failed` left by a previous launch is removed only by rank k itself; a faster peer of the resumed
    launch must not abort on it (markers carry the launch id), and a marker that vanishes while it is read counts as absent."""
    from dg_tta_amd.sharding import _marker_of_this_launch, failed_marker, mark_rank_failed, wait_for_files
    monkeypatch.setenv("DGTTA_LAUNCH_ID", "first")
    mark_rank_failed(tmp_path, 1)
    monkeypatch.setenv("DGTTA_LAUNCH_ID", "second")
    with pytest.raises(TimeoutError):        # not RuntimeError: the stale marker is ignored, the file just never comes
        wait_for_files([tmp_path / "x.pt"], timeout_s=0.3, poll_s=0.05, abort_if=[failed_marker(tmp_path, r) for r in range(2)])
    assert not _marker_of_this_launch(tmp_path / "gone" / ".rank_0.failed", "failed")
    mark_rank_failed(tmp_path, 0)
    with pytest.raises(RuntimeError, match="peer rank failed"):
        wait_for_files([tmp_path / "x.pt"], timeout_s=30, poll_s=0.05, abort_if=[failed_marker(tmp_path, r) for r in range(2)])


def test_done_markers_of_an_earlier_launch_do_not_pass_the_barrier(tmp_path, monkeypatch):
    from dg_tta_amd.sharding import mark_rank_done, wait_for_done_markers
    monkeypatch.setenv("DGTTA_LAUNCH_ID", "first")
    mark_rank_done(tmp_path, 0)
    mark_rank_done(tmp_path, 1)
    wait_for_done_markers(tmp_path, 2, timeout_s=5, poll_s=0.05)
    monkeypatch.setenv("DGTTA_LAUNCH_ID", "second")        # a resumed run in the same directory
    mark_rank_done(tmp_path, 0)
    with pytest.raises(TimeoutError, match="earlier launch"):
        wait_for_done_markers(tmp_path, 2, timeout_s=0.3, poll_s=0.05)
    mark_rank_done(tmp_path, 1)
    wait_for_done_markers(tmp_path, 2, timeout_s=5, poll_s=0.05)


def test_launch_id_is_unique_per_launch_under_torchrun(monkeypatch):
    """torch.distributed.run sets TORCHELASTIC_RUN_ID to the constant 'none' unless --rdzv-id is given; the launch
    id therefore carries the identity (pid, start tick) of the elastic agent, the common parent of one launch's ranks.
    Children of ONE parent agree; children of two different parents differ; an explicit DGTTA_LAUNCH_ID wins; in-process use
    without any launcher keeps the empty id (the sequential-rank tests rely on it)."""
    import subprocess
    import sys
    from dg_tta_amd import sharding
    monkeypatch.delenv("DGTTA_LAUNCH_ID", raising=False)
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    assert sharding.launch_id() == ""
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    here = sharding.launch_id()
    assert here.startswith("none@") and here == sharding.launch_id()
    code = "import sys; sys.path.insert(0, %r); from dg_tta_amd import sharding; print(sharding.launch_id())" % str(__import__("pathlib").Path(__file__).resolve().parents[1])
    env = dict(os.environ, TORCHELASTIC_RUN_ID="none")
    env.pop("DGTTA_LAUNCH_ID", None)
    kids = [subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip()
            for _ in range(2)]
    assert kids[0] == kids[1] and kids[0].startswith("none@")          # two ranks of this "agent" (this process)
    # another launch = another agent process: a shell in between plays the second agent
    other = subprocess.run(["sh", "-c", f"{sys.executable} -c {code!r}; true"], env=env, capture_output=True, text=True,
                           check=True).stdout.strip()
    assert other.startswith("none@") and other != kids[0]
    monkeypatch.setenv("DGTTA_LAUNCH_ID", "explicit")
    assert sharding.launch_id() == "explicit"


def test_launch_id_across_nodes_and_restarts(monkeypatch):
    """the elastic agent's identity is common to ONE node's ranks only.  Two simulated agents (two parents) of a
    two-node job agree on the id when a rendezvous id is given and are refused when it is the constant 'none'; an elastic
    restart (same agent, same id) is a new launch: a failed marker of attempt 0 does not count for attempt 1."""
    import subprocess
    import sys
    from dg_tta_amd import sharding
    root = str(__import__("pathlib").Path(__file__).resolve().parents[1])
    code = "import sys; sys.path.insert(0, %r); from dg_tta_amd import sharding; print(sharding.launch_id())" % root
    base = {k: v for k, v in os.environ.items() if k not in ("DGTTA_LAUNCH_ID", "TORCHELASTIC_RESTART_COUNT")}

    def rank_under_agent(env):          # a shell in between = another parent process = another node's agent
        return subprocess.run(["sh", "-c", f"{sys.executable} -c {code!r}; true"], env=env, capture_output=True, text=True)
    two_nodes = dict(base, TORCHELASTIC_RUN_ID="job-17", WORLD_SIZE="4", LOCAL_WORLD_SIZE="2")
    ids = [rank_under_agent(two_nodes).stdout.strip() for _ in range(2)]
    assert ids[0] == ids[1] == "job-17#0"
    anon = rank_under_agent(dict(two_nodes, TORCHELASTIC_RUN_ID="none"))
    assert anon.stdout.strip() == "" and "--rdzv-id" in anon.stderr
    for var in ("DGTTA_LAUNCH_ID", "TORCHELASTIC_RESTART_COUNT"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    with pytest.raises(RuntimeError, match="rdzv-id"):
        sharding.launch_id()
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")             # one node: the agent's identity + the attempt
    first = sharding.launch_id()
    assert first.startswith("none@") and first.endswith("#0")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    assert sharding.launch_id() == first[:-1] + "1"


def test_bench_eight_ranks_under_torch_distributed_run():
    """The driver's N = 8 launch, rehearsed on the CPU: `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher; stub runner, gloo): one line from rank 0 with eight rates,
    the whole-job value quoted on the slowest rank."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(root / "bench.py"), "--gpus", "8", "--stub-runner", "0.05", "--steps", "2",
                          "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["world_size_seen"] == 8 and len(out["per_rank_epochs_per_s"]) == 8
    assert out["config"]["launcher"] == "torch.distributed.run" and out["scaling"] == "weak"
    assert out["value"] == pytest.approx(8 * out["value_per_gpu"], rel=1e-4)
    assert out["value_per_gpu"] <= min(out["per_rank_epochs_per_s"]) * 1.0001
