"""Host-side logic of the product package (no GPU): snippet/window index plan, result assembly,
multi-process sharding + gather over gloo (world_size 2), deterministic generators."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from mimamo_net_amd import sampler, weights, synthetic, dist as mdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_snippet_ranges_and_window_ids_match_reference(golden):
    g = golden("sampler")
    for n in (10, 64, 100, 128, 309):
        r = sampler.snippet_ranges(n, 64, 64)
        np.testing.assert_array_equal(np.array(r), g["ranges_%d" % n])
        ids = np.stack([sampler.window_ids(s, e, n) for s, e in r])
        assert ids.dtype == np.int32
        np.testing.assert_array_equal(ids % 251, g["ids_%d" % n])
    with pytest.raises(ValueError):
        sampler.snippet_ranges(0)
    assert sampler.snippet_ranges(1) == [[0, 1]]
    np.testing.assert_array_equal(sampler.window_ids(0, 1, 1), np.zeros((1, 13), dtype=np.int32))


def test_product_sampler_equals_oracle(oracle):
    for n in (1, 5, 63, 64, 65, 127, 200, 1000):
        assert sampler.snippet_ranges(n) == oracle.snippet_ranges(n)
        for s, e in sampler.snippet_ranges(n):
            np.testing.assert_array_equal(sampler.window_ids(s, e, n), oracle.window_ids(s, e, n))


def test_assemble_overwrite_order_and_coverage():
    r = sampler.snippet_ranges(150)
    assert r == [[0, 64], [64, 128], [86, 150]]
    preds = [np.full((64, 2), k + 1.0) for k in range(3)]
    v = sampler.assemble(preds, r)
    assert v.shape == (150, 2) and v.dtype == np.float64
    assert (v[:64] == 1).all() and (v[64:86] == 2).all() and (v[86:] == 3).all()
    # like the reference (api/tester.py:112-118, min_f starts at 0) an uncovered prefix is NOT detected:
    v2 = sampler.assemble(preds[1:], r[1:])
    assert (v2[:64] == 0).all()


def test_generators_are_deterministic_and_layouts_complete():
    a = weights.det_uniform("x", (7, 3), -1, 1, 5)
    b = weights.det_uniform("x", (7, 3), -1, 1, 5)
    np.testing.assert_array_equal(a, b)
    assert abs(float(weights.det_uniform("y", (100000,), -1, 1, 1).mean())) < 0.01
    sd = weights.make_two_stream_state_dict(1)
    assert len(sd) == 107 and sum(v.size for k, v in sd.items() if v.dtype == np.float32) == 2636425 + sum(
        sd[k + s].size for k in weights.TWO_STREAM_BN_KEYS for s in (".running_mean", ".running_var"))
    assert weights.two_stream_blob(sd).size == 2640783
    rs = weights.make_resnet50_state_dict(1)
    assert len(weights.resnet50_layers()) == 53 and weights.resnet50_blob(rs).size == 23561152
    c1, c2 = synthetic.make_clip_u8(3, 4), synthetic.make_clip_u8(3, 4)
    np.testing.assert_array_equal(c1, c2)
    assert c1.std() > 10  # textured, never constant


def test_shard_policies():
    assert mdist.shard(10, 0, 4) == [0, 4, 8] and mdist.shard(10, 3, 4) == [3, 7]
    lengths = [300, 64, 64, 64, 200, 100]
    parts = [mdist.shard(6, r, 2, lengths) for r in range(2)]
    assert sorted(parts[0] + parts[1]) == list(range(6))
    loads = [sum(lengths[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 100


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import mimamo_net_amd  # noqa: F401
    from mimamo_net_amd import dist as md
    r, w, _ = md.init("gloo")
    lengths = [64, 100, 37, 64, 128]

    def compute(idx):  # stand-in for the per-video hot path: rows depend only on (video, frame)
        rows = [np.stack([np.full(lengths[i], i, dtype=np.float32), np.arange(lengths[i], dtype=np.float32)], 1) for i in idx]
        return torch.from_numpy(np.concatenate(rows)) if rows else torch.zeros((0, 2))

    res = md.run_sharded(lengths, compute, r, w)
    ok = all(res[i].shape == (lengths[i], 2) and (res[i][:, 0] == i).all() and (res[i][:, 1] == np.arange(lengths[i])).all()
             for i in range(len(lengths)))
    q.put((rank, ok, sorted(res)))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_gloo_shard_and_gather():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in got) and all(keys == [0, 1, 2, 3, 4] for _, _, keys in got)
