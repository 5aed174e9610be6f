"""Pinned host memory for the offload path, placed where the link is.

The reference keeps the target's layers in `pin_memory()` tensors (umbrella/models/llama.py:203-212, llama_layer.py:220-258)
and lets the allocator decide where the pages live.  On a two-socket host that decides the copy rate: a slab on the far
socket crosses the inter-socket fabric on its way to the GPU (round 4 measured 330 - 377 ms per step for the same 40
streamed layers, by what the process had allocated before).  Here the slabs of a model come out of ONE arena:

  * anonymous mmap, transparent huge pages requested (madvise), bound to the NUMA node of the GPU's PCIe root
    (libnuma `numa_tonode_memory` = mbind MPOL_BIND; the node is read from sysfs), touched once, then pinned with ONE
    hipHostRegister -- one registration instead of one per layer;
  * `reserve()` lets a process claim the arena EARLY (bench.py does, before the headline's allocations fragment the
    host); a model that streams takes its slabs from the reserved arena when it fits, else from a fresh one;
  * every step that is not available (no libnuma, a single-node host, no sysfs entry) degrades to the next best thing
    and `describe()` says which: the bench line reports `numa_node`, `arena` and the per-copy rates.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch

_libc = C.CDLL(None, use_errno=True)
_libc.mmap.restype = C.c_void_p
_libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
_libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
_libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
PROT_RW, MAP_PRIVATE_ANON, MADV_HUGEPAGE = 0x1 | 0x2, 0x02 | 0x20, 14
ALIGN = 2 << 20


def gpu_numa_node(device) -> int:
    """NUMA node of the GPU's PCIe function (sysfs), -1 if the host does not say"""
    try:
        p = torch.cuda.get_device_properties(torch.device(device))
        dom, bus, dev = getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id
        with open(f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{dev:02x}.0/numa_node") as f:
            return int(f.read().strip())
    except Exception:
        return -1


def host_nodes() -> int:
    try:
        return len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        return 1


class HostArena:
    def __init__(self, nbytes: int, device):
        self.size = (int(nbytes) + ALIGN - 1) // ALIGN * ALIGN
        self.device = torch.device(device)
        self.node = gpu_numa_node(device)
        if os.environ.get("UMB_HOST_ARENA_NODE"):                        # experiments: place the arena on a chosen node
            self.node = int(os.environ["UMB_HOST_ARENA_NODE"])
        self.nodes = host_nodes()
        self.notes = []
        base = _libc.mmap(None, self.size, PROT_RW, MAP_PRIVATE_ANON, -1, 0)
        if base in (None, C.c_void_p(-1).value):
            raise MemoryError(f"mmap of {self.size} bytes failed (errno {C.get_errno()})")
        self.base = base
        if _libc.madvise(base, self.size, MADV_HUGEPAGE) != 0:
            self.notes.append("no transparent huge pages")
        self.bound = False
        if self.node >= 0 and self.nodes > 1:
            try:
                numa = C.CDLL("libnuma.so.1", use_errno=True)
                numa.numa_tonode_memory.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
                if numa.numa_available() >= 0:
                    # PREFERRED, not strict: numa_tonode_memory binds under libnuma's current bind policy, and under a strict
                    # bind a node without `size` free bytes sends the first-touch memset below to the OOM killer instead of
                    # to this constructor's caller.  Bind only when the node really has the room, and say what happened.
                    numa.numa_set_bind_policy.argtypes = [C.c_int]
                    numa.numa_set_bind_policy(0)
                    numa.numa_node_size64.argtypes = [C.c_int, C.POINTER(C.c_longlong)]
                    numa.numa_node_size64.restype = C.c_longlong
                    free = C.c_longlong(0)
                    total = numa.numa_node_size64(self.node, C.byref(free))
                    if total > 0 and free.value >= self.size + (1 << 30):
                        C.set_errno(0)
                        numa.numa_tonode_memory(base, self.size, self.node)      # (void in libnuma; failures surface in errno)
                        if C.get_errno() == 0:
                            self.bound = True
                        else:
                            self.notes.append(f"numa_tonode_memory errno {C.get_errno()}: first-touch placement")
                    else:
                        self.notes.append(f"node {self.node} has {free.value >> 20} MiB free of the {self.size >> 20} MiB wanted: "
                                          "first-touch placement")
            except OSError:
                self.notes.append("libnuma missing: first-touch placement")
        elif self.nodes <= 1:
            self.notes.append("single NUMA node")
        else:
            self.notes.append("GPU NUMA node unknown")
        C.memset(base, 0, self.size)                                     # first touch under the binding
        rc = int(torch.cuda.cudart().cudaHostRegister(base, self.size, 0))
        if rc != 0:
            _libc.munmap(base, self.size)
            raise RuntimeError(f"hipHostRegister failed ({rc})")
        self.registered = True
        self.off = 0
        self.live = 0

    def alloc(self, nbytes: int):
        """a uint8 CPU tensor of nbytes inside the arena (2 MiB aligned), or None if it does not fit"""
        n = (int(nbytes) + ALIGN - 1) // ALIGN * ALIGN
        if self.off + n > self.size:
            return None
        buf = (C.c_uint8 * int(nbytes)).from_address(self.base + self.off)
        t = torch.frombuffer(buf, dtype=torch.uint8)
        t._arena = self                                                  # the arena outlives its tensors
        self.off += n
        self.live += 1
        weakref.finalize(t, self._gone)
        return t

    def _gone(self):
        self.live = max(0, self.live - 1)

    def release_all(self):
        """forget every slab handed out (their tensors must be gone): the arena can be carved again"""
        self.off, self.live = 0, 0

    def describe(self) -> dict:
        return {"arena": f"mmap {self.size >> 20} MiB, one hipHostRegister" + ("; " + "; ".join(self.notes) if self.notes else ""),
                "numa_node": self.node, "host_numa_nodes": self.nodes, "bound_to_gpu_node": self.bound}

    def close(self):
        if getattr(self, "registered", False):
            torch.cuda.cudart().cudaHostUnregister(self.base)
            self.registered = False
        if getattr(self, "base", None):
            _libc.munmap(self.base, self.size)
            self.base = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_reserved: HostArena | None = None


def reserve(nbytes: int, device) -> HostArena | None:
    """claim an arena now for a model that will stream later (None, with the reason logged by the caller, if it cannot be had)"""
    global _reserved
    if _reserved is not None and _reserved.size >= nbytes:
        return _reserved
    try:
        _reserved = HostArena(nbytes, device)
    except Exception:
        _reserved = None
    return _reserved


def take(nbytes: int, device) -> HostArena | None:
    """the reserved arena if it is idle and large enough, else a new one, else None (the caller falls back to pin_memory)"""
    if os.environ.get("UMB_HOST_ARENA", "1") == "0":
        return None
    if _reserved is not None and _reserved.live == 0 and _reserved.size >= nbytes and _reserved.device == torch.device(device):
        _reserved.release_all()
        return _reserved
    try:
        return HostArena(nbytes, device)
    except Exception:
        return None
