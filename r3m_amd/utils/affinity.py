"""Host-side placement of one rank per GPU: pin the rank's threads to cores of the NUMA node its GPU hangs off.

The reference runs ONE process with nn.DataParallel (/root/reference/r3m/train_representation.py:27-31) and leaves placement
to the OS. Here every GPU has its own process whose main thread enqueues ~1000 kernel launches per step; round 3 measured that
foreign threads next to that launcher cost up to +85 ms on a 94 ms step, and an 8-GPU MI355X node has two sockets: a
launcher scheduled on the far socket pays a cross-socket hop on every doorbell write and every pinned-buffer copy. So each
rank gets a DISJOINT slice of the cores of its GPU's NUMA node (sysfs: /sys/bus/pci/devices/<bdf>/numa_node and
/sys/devices/system/node/node<k>/cpulist); without topology information the process's current CPU set is split evenly.

Pure planning functions (parse_cpulist, plan_rank_cpus) are separated from the sysfs / sched_setaffinity calls so that the
policy is testable on a GPU-less host (tests/test_bench_launch.py)."""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format); '' -> []."""
    cpus = []
    for part in (text or "").strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def plan_rank_cpus(local_rank, gpu_nodes, node_cpus, allowed, max_cpus=16):
    """CPUs for `local_rank`.

    gpu_nodes: NUMA node per LOCAL rank (None / negative = unknown), node_cpus: {node: [cpus]}, allowed: the CPUs this process
    may use at all (its current affinity mask). Ranks that share a node split that node's allowed cpus into equal, disjoint,
    contiguous slices (at most `max_cpus` each: the step needs a launcher, autograd's thread, RCCL's proxy and a loader or two —
    not a socket); a rank with unknown topology takes an even slice of `allowed` by local rank. Never returns an empty list."""
    allowed = sorted(set(allowed))
    world = len(gpu_nodes)
    node = gpu_nodes[local_rank] if 0 <= local_rank < world else None
    pool, peers = None, None
    if node is not None and node >= 0 and node in node_cpus:
        pool = [c for c in node_cpus[node] if c in set(allowed)]
        peers = [r for r in range(world) if gpu_nodes[r] == node]
    if not pool:                                        # unknown node, or the node's cores are outside our cgroup mask
        pool, peers = allowed, list(range(max(world, 1)))
        if local_rank not in peers:
            peers = [local_rank]
    k = peers.index(local_rank)
    per = max(1, len(pool) // len(peers))
    mine = pool[k * per:(k + 1) * per] or pool[-1:]
    return mine[:max_cpus]


def gpu_numa_node(device_index):
    """NUMA node of a visible GPU from sysfs, or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            n = int(f.read().strip())
        return n if n >= 0 else None
    except Exception:
        return None


def numa_cpus():
    """{node: [cpus]} from /sys/devices/system/node."""
    out = {}
    base = "/sys/devices/system/node"
    try:
        for name in os.listdir(base):
            if name.startswith("node") and name[4:].isdigit():
                with open(os.path.join(base, name, "cpulist")) as f:
                    out[int(name[4:])] = parse_cpulist(f.read())
    except OSError:
        pass
    return out


def bind_rank(local_rank, local_world, device_indices=None, max_cpus=16, set_threads=True, min_cpus=1):
    """Pin the CALLING thread — and every thread / process created from it afterwards, which inherit its mask — and cap torch's
    intra-op pool. sched_setaffinity(0, ...) does not move threads that already exist: call this BEFORE init_process_group (RCCL's
    proxy / watchdog threads) and before the loader workers are forked, or those keep the old mask.
    max_cpus: upper bound of the slice (None: the rank's whole share of its NUMA node — a training run with loader workers);
    min_cpus: what the caller needs at least (launcher + autograd thread + its loader workers); a smaller slice (a cgroup-limited
    host: 32 CPUs over 8 ranks) would squeeze the workers next to the launcher — exactly the slowdown the binding is meant to
    avoid — so the mask is then left alone and the returned dict says so.
    Returns a dict describing what was done — bench.py prints it per rank. Never raises: placement is an optimisation."""
    info = {"local_rank": local_rank, "numa_node": None, "cpus": None, "threads": None}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        devs = list(device_indices) if device_indices is not None else list(range(local_world))
        nodes = [gpu_numa_node(d) for d in devs]
        cpus = plan_rank_cpus(local_rank, nodes, numa_cpus(), allowed, max_cpus=max_cpus if max_cpus else len(allowed))
        info["numa_node"] = nodes[local_rank] if local_rank < len(nodes) else None
        if len(cpus) < min_cpus:
            info["skipped"] = f"slice of {len(cpus)} cpu(s) < {min_cpus} needed: affinity left as it was"
            return info
        os.sched_setaffinity(0, cpus)
        info["cpus"] = f"{cpus[0]}-{cpus[-1]}" if cpus == list(range(cpus[0], cpus[-1] + 1)) else ",".join(map(str, cpus))
        if set_threads:
            import torch
            n = max(1, min(4, len(cpus)))
            torch.set_num_threads(n)
            info["threads"] = n
    except Exception as e:   # e.g. no sched_setaffinity on this platform / restricted container
        info["error"] = f"{type(e).__name__}: {e}"
    return info
