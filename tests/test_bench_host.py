"""Host-side pieces of bench.py that need no GPU: the synthetic data kinds and the bookkeeping that ties the static PMC
figures to the sources they were captured on."""
import json
import os

import numpy as np

import bench


def test_data_kinds_have_the_documented_structure():
    n, d, q = 20000, 24, 150
    base = bench.gen_mslr_shaped(3, n, d, q)
    assert base[0].shape == (n, d) and base[0].dtype == np.float32 and len(np.unique(base[2])) == q
    hard = bench.gen_mslr_shaped(3, n, d, q, "hard")
    assert np.array_equal(hard[1], base[1]) and not np.array_equal(hard[0][:, 0], base[0][:, 0])  # weaker label signal
    for kind in ("ties", "tiesmix"):
        X, y, qid = bench.gen_mslr_shaped(3, n, d, q, kind)
        assert (X == np.floor(X)).all(), "every column quantised"
        # duplicated feature rows inside a query, and (tiesmix only) some of them with different labels
        dup_same = dup_diff = 0
        for qq in np.unique(qid)[:40]:
            rows = np.nonzero(qid == qq)[0]
            seen = {}
            for i in rows:
                key = X[i].tobytes()
                if key in seen:
                    if y[seen[key]] == y[i]:
                        dup_same += 1
                    else:
                        dup_diff += 1
                seen.setdefault(key, i)
        assert dup_same > 50
        assert (dup_diff == 0) if kind == "ties" else (dup_diff > 10)


def test_pmc_constants_carry_the_hash_of_their_sources(tmp_path, monkeypatch):
    cur = bench.source_sha1()
    assert set(cur) == set(bench.PMC_SOURCES) and all(v and len(v) == 40 for v in cur.values())
    tj, meta = bench.load_pmc("30k", usable=False)
    assert tj == {} and meta["stale"] is None
    tj, meta = bench.load_pmc("30k", usable=True)
    committed = json.load(open(os.path.join(bench.ROOT, "profiles", "hbm_traffic.json")))["30k"].get("captured", {})
    assert meta["current_sha1"] == cur and meta["captured_sha1"] == committed.get("sha1")
    assert meta["stale"] == (committed.get("sha1") != cur)
    # a capture without hashes (round 1's file) counts as stale
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "fastrank_amd" / "csrc")
    (tmp_path / "profiles" / "hbm_traffic.json").write_text(json.dumps({"30k": {"bench_timed_bytes_per_group": 1.0}}))
    tj, meta = bench.load_pmc("30k", usable=True)
    assert meta["stale"] is True and tj["bench_timed_bytes_per_group"] == 1.0


def test_thread_ranks_exchange_like_process_ranks():
    """bench.py --gpus N from a plain shell runs its ranks as host threads (ThreadComm): barrier, reductions in rank order,
    the gather of restart records and the shared block counter must behave like their torch.distributed counterparts."""
    import threading

    world = 3
    hub = bench.ThreadHub(world)
    hub.devices = [0] * world  # three contexts on one GPU: the exchange stays in host memory, RCCL reports why it did not run
    out = [None] * world

    def body(rank):
        c = bench.ThreadComm(hub, rank)
        s = c.allreduce([float(rank + 1), 0.1 * (rank + 1)], "sum")
        m = c.allreduce([float(rank)], "max")
        rows = c.allgather_rows([float(rank), float(rank * rank)])
        mine = [{"restart_id": r, "score": float(r), "weights": [float(r)]} for r in range(rank, 7, world)]
        allr = c.gather_restarts(mine, 7)
        blocks = list(c.steal_blocks(10, 3))
        c.barrier()
        blocks2 = list(c.steal_blocks(4, 4))  # (a second job gets its own counter)
        out[rank] = (s, m, rows, [r["restart_id"] for r in allr], blocks, blocks2, c.rccl_report)

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(o is not None for o in out)
    for s, m, rows, ids, _, _, rccl in out:
        assert s == [6.0, (0.1 + 0.2) + 0.30000000000000004] and m == [2.0]  # added in rank order: the same bits on every rank
        assert rows == [[0.0, 0.0], [1.0, 1.0], [2.0, 4.0]] and ids == list(range(7))
        assert rccl["ran"] is False and rccl["ranks"] == 3 and "share device 0" in rccl["reason"]
    assert sorted(b for o in out for b in o[4]) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert sorted(b for o in out for b in o[5]) == [(0, 4)]
    single = bench.SingleComm()
    assert single.allreduce([1.0], "sum") == [1.0] and list(single.steal_blocks(5, 2)) == [(0, 2), (2, 4), (4, 5)]


def test_thread_ranks_on_distinct_gpus_must_exchange_through_rccl():
    """with one GPU per rank thread the job's exchange goes through ONE single-process RCCL all-gather, and
    when that cannot run it is an ERROR on every rank, not a silent host-memory gather.  This box has no GPU: two ranks that
    name devices 0 and 1 must both fail loudly."""
    import threading

    from fastrank_amd import native
    if native.device_count() >= 2:
        import pytest
        pytest.skip("needs a box without two GPUs (the multi-GPU form is tests/test_gpu_multidevice.py's)")
    hub = bench.ThreadHub(2)
    hub.devices = [0, 1]
    errs = [None, None]

    def body(rank):
        c = bench.ThreadComm(hub, rank)
        try:
            c.gather_restarts([{"restart_id": rank, "score": 0.5, "weights": [1.0, 2.0]}], 2)
        except RuntimeError as exc:
            errs[rank] = str(exc)

    threads = [threading.Thread(target=body, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(e is not None and "RCCL exchange failed" in e for e in errs), errs


def test_rccl_failure_is_an_error_only_when_every_rank_has_its_own_gpu():
    """`--backend nccl` on a node with one GPU per rank must not turn into a gloo run silently; several ranks
    pinned to ONE GPU (what the one-GPU box can exercise) legitimately fall back."""
    assert bench.rccl_is_mandatory("nccl", 8, 8, None) is True
    assert bench.rccl_is_mandatory("nccl", 2, 8, None) is True
    assert bench.rccl_is_mandatory("nccl", 2, 1, None) is False     # two ranks, one GPU visible
    assert bench.rccl_is_mandatory("nccl", 2, 8, "0") is False      # FR_BENCH_DEVICE pins every rank to one ordinal
    assert bench.rccl_is_mandatory("gloo", 8, 8, None) is False
    assert bench.rccl_is_mandatory("nccl", 1, 8, None) is False


def test_peer_copy_report_covers_every_other_device_and_survives_errors():
    class FakeNative:
        def __init__(self):
            self.calls = []

        def peer_copy(self, src, dst, nbytes):
            self.calls.append((src, dst))
            if dst == 3:
                raise RuntimeError("no such device")
            return {"can_access": True, "enabled": True, "gbps": 48.0, "ms": 5.6}

    fake = FakeNative()
    rep = bench.peer_copy_report(fake, [0, 1, 2, 3])
    assert fake.calls == [(0, 1), (0, 2), (0, 3)]
    assert [r["dst"] for r in rep] == [1, 2, 3] and rep[0]["gbps"] == 48.0 and "error" in rep[2]
    rep = bench.peer_copy_report(FakeNative(), [0, 0])  # two ranks on one GPU: one same-device copy
    assert len(rep) == 1 and rep[0]["same_device"] is True


def test_power_leg_parses_rocm_smi_text_per_device():
    """bench.py's power / clock leg reads `rocm-smi --showpower --showclocks` as text: the figures of the rank's own device,
    nothing when the tool prints something else (the leg then leaves `power` null instead of failing the bench)."""
    import bench
    txt = ("GPU[0]\t\t: fclk clock level: 0: (1250Mhz)\n"
           "GPU[0]\t\t: sclk clock level: 1: (2211Mhz)\n"
           "GPU[0]\t\t: socclk clock level: S: (70Mhz)\n"
           "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1358.0\n"
           "GPU[1]\t\t: sclk clock level: S: (94Mhz)\n"
           "GPU[1]\t\t: Average Graphics Package Power (W): 242.0\n")
    assert bench.parse_rocm_smi(txt, 0) == (1358.0, 2211)
    assert bench.parse_rocm_smi(txt, 1) == (242.0, 94)
    assert bench.parse_rocm_smi(txt, 2) is None
    assert bench.parse_rocm_smi("rocm-smi: command not found", 0) is None
