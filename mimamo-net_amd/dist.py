"""Multi-GPU scaling: videos are independent units, so they shard across ranks with NO collective on the
data path.  One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  The only exchanges are the work-queue broadcast (rank 0 -> all: the list of
videos and their lengths, a few bytes per video) and the gather of [frames, 2] fp32 results (8 bytes per
frame) so every rank -- or just rank 0 -- can assemble the per-video tables.  The atomic unit is one video's
snippet batch (api/tester.py:69-72): a video is never split across ranks.

The reference is single-process / single-device (api/steerable/utils.py:34-50); this module is the build's
addition, not a port of anything.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import torch


def env_world():
    """(rank, world, local_rank) from the torchrun environment; world is None when WORLD_SIZE is unset."""
    w = os.environ.get("WORLD_SIZE")
    return int(os.environ.get("RANK", "0")), (int(w) if w is not None else None), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, device_index=None, force=False):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank).

    backend: "nccl" (RCCL; default when a GPU is present) or "gloo".  device_index: the GPU this rank binds to
    (default LOCAL_RANK).  force: create the process group even for world == 1 (exercises RCCL on a 1-GPU box)."""
    import torch.distributed as dist
    rank, world, local_rank = env_world()
    world = 1 if world is None else world
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        # bind a GPU only where one is asked for: RCCL ranks (one per GPU), or an explicit device_index.  A gloo plumbing run
        # on a GPU host with fewer GPUs than ranks (tests, --stub-compute) must not touch cuda:<local_rank>
        if torch.cuda.is_available() and (backend == "nccl" or device_index is not None):
            torch.cuda.set_device(local_rank if device_index is None else device_index)
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_cpus(device_index):
    """(numa node, CPUs local to it) of GPU `device_index` from sysfs (/sys/bus/pci/devices/<bdf>/{numa_node,local_cpulist}),
    or (None, []) when the platform does not say (no GPU, no sysfs entry, numa_node == -1)."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        base = "/sys/bus/pci/devices/" + bdf
        with open(base + "/numa_node") as f:
            node = int(f.read().strip())
        with open(base + "/local_cpulist") as f:
            cpus = _parse_cpulist(f.read())
        return (node if node >= 0 else None), (cpus if node >= 0 else [])
    except Exception:
        return None, []


def bind_rank_cpus(local_rank, local_world, device_of_rank=None):
    """Pin this rank's host threads next to its GPU.  device_of_rank[r] = GPU index of local rank r (default: r; all zeros under
    bench.py --same-device).  With NUMA information: the rank's share of the CPUs of its GPU's node -- the ranks whose GPUs sit on
    the same node split that node's CPUs (hyper-thread siblings included) in rank order.  Without (no sysfs entry, numa_node ==
    -1, no GPU): an even contiguous slice of the CPUs this process may already use.  Returns {"numa_node", "cpus": count,
    "first_cpu", "bound"} for the per-rank diagnostics; never raises (an unsupported platform simply stays unbound).

    The reference is single-process (api/steerable/utils.py:34-50): nothing to mirror, this is the build's own layer."""
    info = {"numa_node": None, "cpus": 0, "first_cpu": -1, "bound": False}
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return info
    info["cpus"], info["first_cpu"] = len(allowed), (allowed[0] if allowed else -1)
    if local_world <= 1 or not allowed:
        return info
    node, local, peers = None, [], list(range(local_world))
    if device_of_rank is not None and torch.cuda.is_available():
        nodes = [gpu_numa_cpus(d)[0] for d in device_of_rank]
        node, local = gpu_numa_cpus(device_of_rank[local_rank])
        local = [c for c in local if c in set(allowed)]
        if node is not None and local:
            peers = [r for r in range(local_world) if nodes[r] == node]
    pool = local if (node is not None and local) else allowed
    k, i = len(peers), peers.index(local_rank)
    per = max(1, len(pool) // k)
    mine = pool[i * per:(i + 1) * per] if i < k - 1 else pool[i * per:]
    if not mine:
        return info
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return info
    info.update({"numa_node": node if pool is local else None, "cpus": len(mine), "first_cpu": mine[0], "bound": True})
    return info


def all_gather_bytes(local, world, device="cpu"):
    """Equal-shape all-gather of a uint8 tensor [n, ...] -> [world * n, ...] on `device` (RCCL: device to device).  Used to
    assemble the synthetic clip pool: every rank generates 1/world of the distinct clip contents, nobody generates one twice."""
    if not active():
        return local.to(device)
    import torch.distributed as dist
    t = local.to(torch.device(device)).contiguous()
    if _staged(t):
        t = t.cpu()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather(list(out.chunk(world, 0)), t)      # views of `out`: every rank's piece lands in place
    return out


def active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def _staged(t):
    """gloo has no device all_gather/barrier: stage device tensors through the host for it (tests only; RCCL moves
    them device to device)."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() == "gloo"


def shard(n_videos, rank, world, lengths=None):
    """Indices of the videos rank `rank` processes.  Equal-length clips: round-robin (c mod world == rank).
    With `lengths`: longest-first greedy onto the least-loaded rank (deterministic, identical on all ranks)."""
    if lengths is None:
        return list(range(rank, n_videos, world))
    order = sorted(range(n_videos), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += int(lengths[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def broadcast_work(work, rank, world, device="cpu"):
    """The work queue: rank 0 holds `work` = int64 array [n, k] (one row per video: id, length, ...); every rank
    returns the same array.  Two broadcasts (shape, payload) from rank 0; on RCCL they run device to device."""
    if world == 1 and not active():
        return np.ascontiguousarray(np.asarray(work, dtype=np.int64))
    import torch.distributed as dist
    dev = torch.device(device)
    if rank == 0:
        w = np.ascontiguousarray(np.asarray(work, dtype=np.int64))
        assert w.ndim == 2
        shape = torch.tensor(list(w.shape), dtype=torch.int64, device=dev)
    else:
        shape = torch.zeros(2, dtype=torch.int64, device=dev)
    dist.broadcast(shape, src=0)
    n, k = int(shape[0].item()), int(shape[1].item())
    payload = torch.from_numpy(w).to(dev) if rank == 0 else torch.empty((n, k), dtype=torch.int64, device=dev)
    dist.broadcast(payload, src=0)
    return payload.cpu().numpy()


def all_gather_rows_async(local_rows, world):
    """Equal-shape all-gather of [n, C] rows, asynchronous: returns (work handle or None, list of per-rank tensors).
    The caller waits on the handle before reading the list (bench: one step later, so a rank never stalls on a
    slower peer inside a step)."""
    if not active():
        return None, [local_rows]
    import torch.distributed as dist
    if _staged(local_rows):
        host = local_rows.cpu()
        bufs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(bufs, host)
        return None, bufs
    bufs = [torch.empty_like(local_rows) for _ in range(world)]
    return dist.all_gather(bufs, local_rows, async_op=True), bufs


def gather_rows(local_rows, world, rank, max_rows=None):
    """All-gather variable-length [n_r, C] float tensors; returns the list of per-rank tensors (on every rank)."""
    if world == 1 and not active():
        return [local_rows]
    import torch.distributed as dist
    if _staged(local_rows):
        local_rows = local_rows.cpu()
    n = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=local_rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(counts) if max_rows is None else max_rows
    pad = torch.zeros((cap, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [b[:c] for b, c in zip(bufs, counts)]


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of one float (the job time is the slowest rank's)."""
    if not active():
        return float(value)
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=torch.device(device))
    if _staged(t):
        t = t.cpu()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_floats(values, device="cpu"):
    """Every rank contributes the same number of floats; returns a [world, k] float64 numpy array on every rank
    (diagnostics: per-rank step times, frame counts)."""
    v = np.asarray(values, dtype=np.float64).reshape(-1)
    if not active():
        return v[None, :].copy()
    import torch.distributed as dist
    t = torch.from_numpy(v.copy()).to(torch.device(device))
    if _staged(t):
        t = t.cpu()
    bufs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, t)
    return torch.stack([b.cpu() for b in bufs], 0).numpy()


def barrier():
    if active():
        import torch.distributed as dist
        dist.barrier()


def shutdown():
    if active():
        import torch.distributed as dist
        dist.destroy_process_group()


def run_sharded(video_lengths, compute_rows, rank, world, device="cpu"):
    """Shard videos, run `compute_rows(video_indices) -> [sum(len), C] tensor` locally, gather, and return
    {video index: [len, C] numpy array} for ALL videos on every rank."""
    mine = shard(len(video_lengths), rank, world, video_lengths)
    rows = compute_rows(mine)
    parts = gather_rows(rows.to(device), world, rank)
    out = {}
    for r, part in enumerate(parts):
        part = part.cpu().numpy()
        off = 0
        for i in shard(len(video_lengths), r, world, video_lengths):
            out[i] = part[off: off + video_lengths[i]]
            off += video_lengths[i]
        assert off == part.shape[0]
    return out


# ---- launcher: one process per GPU -------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(argv, n_ranks, env=None, timeout=None):
    """Start `n_ranks` copies of `python argv...` on this node, one per GPU, with the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), wait for all of them and return
    (exit code, rank 0's stdout).  If a rank fails, the others (possibly blocked in a collective) are terminated
    by PID and the failing rank's stderr tail is written to this process's stderr."""
    import tempfile
    import time
    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(n_ranks), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
                 "HSA_ENABLE_IPC_MODE_LEGACY": base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                 # a hardware queue per stream (lanes + default + copy + RCCL's): see bench.py
                 "GPU_MAX_HW_QUEUES": base.get("GPU_MAX_HW_QUEUES", "8")})
    tmp = tempfile.mkdtemp(prefix="mm_ranks_")
    procs, files = [], []
    for r in range(n_ranks):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        fo, fe = open(os.path.join(tmp, "out%d" % r), "w+"), open(os.path.join(tmp, "err%d" % r), "w+")
        files.append((fo, fe))
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, stdout=fo, stderr=fe))
    code, t0 = 0, time.time()
    try:
        while True:
            states = [p.poll() for p in procs]
            bad = [r for r, s in enumerate(states) if s not in (None, 0)]
            if bad:
                code = states[bad[0]]
                files[bad[0]][1].seek(0)
                sys.stderr.write("[rank %d exited with %d]\n%s\n" % (bad[0], code, files[bad[0]][1].read()[-4000:]))
                break
            if all(s == 0 for s in states):
                break
            if timeout is not None and time.time() - t0 > timeout:
                code = 124
                sys.stderr.write("[spawn_ranks: timeout after %.0f s]\n" % timeout)
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    files[0][0].seek(0)
    out0 = files[0][0].read()
    for fo, fe in files:
        fo.close()
        fe.close()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return code, out0
