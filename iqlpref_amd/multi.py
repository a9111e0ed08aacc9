"""Several independent IQL seeds on ONE GPU.

One seed at batch 256 is a latency chain that leaves most of an MI355X idle;
the reference runs several W&B agents per GPU for the same reason
(``ensemble_sweeps/launch.sh:12`` AGENTS_PER_GPU).  ``SeedGroup`` is that, inside one process.
The trainers share nothing (own arenas, own Philox stream) and the arithmetic of every seed
is bit-identical to running it alone.  Two execution modes:

``mode="group"``
    ONE launch sequence steps all seeds: every kernel of the step runs with gridDim.y = K
    (``iqlhip_group_train_steps``).  K x the work-groups per launch, one third of the kernel
    boundaries per seed-step.
``mode="streams"``
    every trainer replays its own hipGraph on its own HIP stream; the launches interleave.
``mode="split"`` (``n_streams=G``, default 2; the default mode for trainers of one shape)
    G sub-groups, each stepped by its own launch sequence on its own HIP stream, every stream
    confined to its own slice of the compute units (bit i of the CU mask belongs to slice i % G,
    ``iqlhip_stream_create_cu_slice``; the mask runs round-robin over the 8 XCDs, so two slices
    are the even and the odd XCDs).  The sub-groups share no CU and no L2, only the memory system:
    one's HBM-bound k_update runs beside the other's latency-bound k_forward / k_backward.
    Measured (tools/group_streams.py, tools/group_scan.py): 2 / 4 / 8 / 16 seeds as two sub-groups
    107k / 160k / 202k / 250k steps/s against 93k / 130k / 171k / 204k as one group.  Two slices are the
    sweet spot (three do not divide the chip's 8 XCDs evenly, four leave each sub-group too few CUs).
"""
import ctypes as C
from typing import List, Optional, Sequence, Union

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .iql import ImplicitQLearning, ReplayBuffer


# The CU-slice streams of a device are created once per process and shared by every split-mode
# group: a slice stream created after an earlier one was destroyed landed on the hardware queue
# of its sibling (both halves of the chip then serialised: 91k instead of 190k steps/s).
_SLICE_STREAMS = {}


def _slice_streams(lib, dev, n_slices: int):
    key = (torch.device(dev).index or 0, n_slices)
    if key not in _SLICE_STREAMS:
        raw = []
        with torch.cuda.device(dev):
            for g in range(n_slices):
                st = C.c_void_p()
                check(lib.iqlhip_stream_create_cu_slice(C.byref(st), g, n_slices))
                raw.append(st)
        _SLICE_STREAMS[key] = (raw, [torch.cuda.ExternalStream(st.value, device=dev) for st in raw])
    return list(_SLICE_STREAMS[key][1])


def _shape_key(t: ImplicitQLearning):
    return (t._state_dim, t._action_dim, t._hidden, t._n_hidden, t._precision, t._deterministic, t._n_critics,
            bool(t._dropout))


def _on_tuned_step(t: ImplicitQLearning) -> bool:
    """Group launches exist for the tuned three-kernel step only (include/iqlhip.h, shape envelope);
    trainers of other shapes (n_hidden != 2, other widths) are stepped one by one, each on its stream."""
    return t._n_hidden == 2 and t._hidden in (64, 128, 256)


class SeedGroup:
    def __init__(self, trainers: Sequence[ImplicitQLearning], chunk: int = 2000, mode: Optional[str] = None,
                 n_streams: int = 2):
        if not trainers:
            raise ValueError("SeedGroup needs at least one trainer")
        devs = {t._dev for t in trainers}
        if len(devs) != 1:
            raise ValueError("all trainers of a SeedGroup must live on one device")
        if len({id(t) for t in trainers}) != len(trainers):
            raise ValueError("a trainer may appear only once in a SeedGroup")
        one_shape = len({_shape_key(t) for t in trainers}) == 1 and len(trainers) <= _lib.MAX_GROUP and \
            all(_on_tuned_step(t) for t in trainers)
        if mode is None:  # the fastest arrangement measured for the shape at hand
            mode = ("split" if len(trainers) >= 2 else "group") if one_shape else "streams"
        if mode not in ("group", "streams", "split"):
            raise ValueError("mode must be 'group', 'streams' or 'split'")
        if mode in ("group", "split") and not one_shape:
            raise ValueError(f"mode='group' needs at most {_lib.MAX_GROUP} trainers of one shape (dims, hidden, "
                             "precision, policy kind, critics, dropout on/off) that runs on the tuned step "
                             "(n_hidden = 2, hidden_dim 64 / 128 / 256)")
        self.mode = mode
        self.trainers: List[ImplicitQLearning] = list(trainers)
        self._dev = next(iter(devs))
        self._lib = _lib.load()
        self._streams = [torch.cuda.Stream(device=self._dev) for _ in self.trainers] if mode == "streams" else []
        self._chunk = int(chunk)
        self._group = None
        self._group_batch = None
        self._children: List["SeedGroup"] = []
        self._child_slices: List[slice] = []
        if mode == "split":
            G = max(1, min(int(n_streams), len(self.trainers)))
            base, extra, lo = len(self.trainers) // G, len(self.trainers) % G, 0
            for g in range(G):  # contiguous runs of trainers, sizes differing by at most one
                hi = lo + base + (1 if g < extra else 0)
                self._children.append(SeedGroup(self.trainers[lo:hi], mode="group"))
                self._child_slices.append(slice(lo, hi))
                lo = hi
            self._chunk = min(self._chunk, 500)  # short turns keep the queues of all streams fed

    def __len__(self):
        return len(self.trainers)

    # -- group handle --------------------------------------------------------- #
    def _ensure_group(self, batch_size: int):
        if self._group is not None and self._group_batch == batch_size and \
                all(t._handle is not None and t._handle_batch == batch_size for t in self.trainers):
            return
        self._drop_group()
        for t in self.trainers:
            t._ensure_handle(batch_size)
        arr = (C.c_void_p * len(self.trainers))(*[t._handle.value for t in self.trainers])
        g = C.c_void_p()
        with torch.cuda.device(self._dev):
            check(self._lib.iqlhip_group_create(C.byref(g), arr, len(self.trainers)))
        self._group, self._group_batch = g, batch_size
        for t in self.trainers:
            t._group_owner = self

    def _drop_group(self):
        if getattr(self, "_group", None) is not None:
            self._lib.iqlhip_group_destroy(self._group)
            self._group = None
            for t in self.trainers:
                t._group_owner = None

    def close(self):
        """Dissolve the device-side group; the trainers stay usable on their own."""
        for ch in getattr(self, "_children", []):
            ch.close()
        self._drop_group()
        if getattr(self, "mode", None) == "split":
            for st in self._streams:
                st.synchronize()
            self._streams = []  # (the slice streams themselves are shared and stay, see _slice_streams)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- stepping ------------------------------------------------------------- #
    def train_steps(self, replay: Union[ReplayBuffer, Sequence[ReplayBuffer]], n_steps: int, batch_size: int, *,
                    indices: Optional[Sequence[Optional[torch.Tensor]]] = None,
                    dropout_keep: Optional[Sequence[Optional[torch.Tensor]]] = None,
                    return_losses: bool = False, graph_unroll: Optional[int] = None):
        """``n_steps`` x (sample + train) for every seed.  ``replay`` is one buffer shared by all
        seeds (a sweep varies the seed only) or one per seed; ``indices`` / ``dropout_keep`` are
        optional per-seed lists (entries may be None) with the meaning of
        ``ImplicitQLearning.train_steps``.  Returns a list of [n_steps, 3] loss tensors when
        ``return_losses``.  Asynchronous: call ``synchronize()`` (or read the losses) before
        touching the parameters."""
        K = len(self.trainers)
        bufs = list(replay) if isinstance(replay, (list, tuple)) else [replay] * K
        if len(bufs) != K:
            raise ValueError("one replay buffer per trainer (or a single shared one)")
        idx = list(indices) if indices is not None else [None] * K
        keep = list(dropout_keep) if dropout_keep is not None else [None] * K
        if len(idx) != K or len(keep) != K:
            raise ValueError("indices / dropout_keep: one entry per trainer")
        if self.mode == "streams":
            return self._train_streams(bufs, n_steps, batch_size, idx, keep, return_losses, graph_unroll)
        if self.mode == "split":
            return self._train_split(bufs, n_steps, batch_size, idx, keep, return_losses, graph_unroll)
        self._ensure_group(batch_size)
        for i, t in enumerate(idx):
            if t is not None:
                if t.dtype != torch.int64 or tuple(t.shape) != (n_steps, batch_size):
                    raise ValueError("indices must be int64 [n_steps, batch_size]")
                idx[i] = t.contiguous()
        keep = [None if k is None else k.to(torch.uint8).contiguous() for k in keep]
        losses = [torch.empty((n_steps, 3), dtype=torch.float32, device=self._dev) for _ in range(K)] \
            if return_losses else None
        views = (_lib.ReplayView * K)(*[b.view() for b in bufs])
        parr = lambda ts: (C.c_void_p * K)(*[None if t is None else t.data_ptr() for t in ts])
        any_idx, any_keep = any(t is not None for t in idx), any(t is not None for t in keep)
        for t in self.trainers:
            t._refresh_lrs()
        # (group launches: graphs of 50 steps 202-204k steps/s for 8 seeds, plain launches 198-200k)
        unroll = 50 if graph_unroll is None else graph_unroll
        with torch.cuda.device(self._dev):
            check(self._lib.iqlhip_group_train_steps(
                self._group, views, n_steps, parr(idx) if any_idx else None,
                parr(keep) if any_keep else None, parr(losses) if losses is not None else None,
                unroll, stream_ptr()))
        for t in self.trainers:
            t._after_steps(n_steps)
        return losses

    def _train_streams(self, bufs, n_steps, batch_size, idx, keep, return_losses, graph_unroll):
        cur = torch.cuda.current_stream(self._dev)
        for st in self._streams:
            st.wait_stream(cur)
        out = [[] for _ in self.trainers]
        done = 0
        while done < n_steps:  # round-robin in chunks so that the queues of all streams stay fed
            c = min(self._chunk, n_steps - done)
            for k, (tr, st, buf) in enumerate(zip(self.trainers, self._streams, bufs)):
                with torch.cuda.stream(st):
                    # (one host thread feeds every stream: graphs keep its share per step small)
                    r = tr.train_steps(buf, c, batch_size, return_losses=return_losses,
                                       indices=None if idx[k] is None else idx[k][done:done + c],
                                       dropout_keep=None if keep[k] is None else keep[k][done:done + c],
                                       graph_unroll=8 if graph_unroll is None else graph_unroll)
                    if return_losses:
                        out[k].append(r)
            done += c
        for st in self._streams:
            cur.wait_stream(st)
        if return_losses:
            return [torch.cat(o) for o in out]
        return None

    def _ensure_slice_streams(self):
        if self._streams:
            return
        try:
            self._streams = _slice_streams(self._lib, self._dev, len(self._children))
        except Exception as e:  # no CU-masked streams here: plain streams keep the results, not the speed
            import warnings
            warnings.warn(f"SeedGroup(mode='split'): CU-slice streams unavailable ({e}); using plain streams")
            self._streams = [torch.cuda.Stream(device=self._dev) for _ in self._children]

    def _train_split(self, bufs, n_steps, batch_size, idx, keep, return_losses, graph_unroll):
        self._ensure_slice_streams()
        cur = torch.cuda.current_stream(self._dev)
        for st in self._streams:
            st.wait_stream(cur)
        out = [[] for _ in self.trainers]
        done = 0
        while done < n_steps:
            c = min(self._chunk, n_steps - done)
            for ch, sl, st in zip(self._children, self._child_slices, self._streams):
                cut = lambda ts: [None if t is None else t[done:done + c] for t in ts[sl]]
                with torch.cuda.stream(st):
                    r = ch.train_steps(bufs[sl], c, batch_size, indices=cut(idx), dropout_keep=cut(keep),
                                       return_losses=return_losses, graph_unroll=graph_unroll)
                if return_losses:
                    for k, rk in zip(range(sl.start, sl.stop), r):
                        out[k].append(rk)
            done += c
        for st in self._streams:
            cur.wait_stream(st)
        if return_losses:
            return [torch.cat(o) for o in out]
        return None

    def kernel_times(self, replay, batch_size: int, n_steps: int = 200):
        """Average duration (us) of the forward / backward / update launches of this group
        (mode "group"): ``n_steps`` eager steps with a HIP-event pair around every launch."""
        if self.mode != "group":
            raise ValueError("kernel_times needs mode='group'")
        self._ensure_group(batch_size)
        check(self._lib.iqlhip_group_set_timing(self._group, 1))
        try:
            self.train_steps(replay, n_steps, batch_size, graph_unroll=0)
            avg, n = (C.c_double * 3)(), C.c_int64()
            check(self._lib.iqlhip_group_get_timing(self._group, C.byref(avg), C.byref(n)))
        finally:
            check(self._lib.iqlhip_group_set_timing(self._group, 0))
        return {"k_forward": avg[0] * 1e3, "k_backward": avg[1] * 1e3, "k_update": avg[2] * 1e3,
                "launches": int(n.value)}

    def synchronize(self):
        for st in self._streams:
            st.synchronize()
        torch.cuda.current_stream(self._dev).synchronize()
