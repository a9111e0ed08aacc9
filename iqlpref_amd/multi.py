"""Several independent IQL seeds on ONE GPU, each on its own HIP stream.

One seed at batch 256 is a latency chain that leaves most of an MI355X idle;
the reference runs several W&B agents per GPU for the same reason
(``ensemble_sweeps/launch.sh:12`` AGENTS_PER_GPU).  ``SeedGroup`` is that, inside one process:
the trainers share nothing (own arenas, own Philox stream, own hipGraph), their launches
interleave on separate streams, and the arithmetic of every seed is bit-identical to running
it alone.
"""
from typing import List, Optional, Sequence, Union

import torch

from .iql import ImplicitQLearning, ReplayBuffer


class SeedGroup:
    def __init__(self, trainers: Sequence[ImplicitQLearning], chunk: int = 2000):
        if not trainers:
            raise ValueError("SeedGroup needs at least one trainer")
        devs = {t._dev for t in trainers}
        if len(devs) != 1:
            raise ValueError("all trainers of a SeedGroup must live on one device")
        self.trainers: List[ImplicitQLearning] = list(trainers)
        self._dev = next(iter(devs))
        self._streams = [torch.cuda.Stream(device=self._dev) for _ in self.trainers]
        self._chunk = int(chunk)

    def __len__(self):
        return len(self.trainers)

    def train_steps(self, replay: Union[ReplayBuffer, Sequence[ReplayBuffer]], n_steps: int, batch_size: int, *,
                    return_losses: bool = False, graph_unroll: Optional[int] = None):
        """``n_steps`` x (sample + train) for every seed.  ``replay`` is one buffer shared by all
        seeds (a sweep varies the seed only) or one per seed.  Returns a list of [n_steps, 3]
        loss tensors when ``return_losses``.  Asynchronous: call ``synchronize()`` (or read the
        losses) before touching the parameters."""
        bufs = list(replay) if isinstance(replay, (list, tuple)) else [replay] * len(self.trainers)
        if len(bufs) != len(self.trainers):
            raise ValueError("one replay buffer per trainer (or a single shared one)")
        cur = torch.cuda.current_stream(self._dev)
        for st in self._streams:
            st.wait_stream(cur)
        out = [[] for _ in self.trainers]
        done = 0
        while done < n_steps:  # round-robin in chunks so that the queues of all streams stay fed
            c = min(self._chunk, n_steps - done)
            for k, (tr, st, buf) in enumerate(zip(self.trainers, self._streams, bufs)):
                with torch.cuda.stream(st):
                    r = tr.train_steps(buf, c, batch_size, return_losses=return_losses,
                                       graph_unroll=graph_unroll)
                    if return_losses:
                        out[k].append(r)
            done += c
        for st in self._streams:
            cur.wait_stream(st)
        if return_losses:
            return [torch.cat(o) for o in out]
        return None

    def synchronize(self):
        for st in self._streams:
            st.synchronize()
