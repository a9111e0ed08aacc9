"""TEST INFRASTRUCTURE ONLY -- CPU restatement, never imported by the product path.

Philox4x32-10 counter-based generator (Salmon et al., SC'11; the same round
function torch's CUDA generator uses) and the index / dropout streams the HIP
path derives from it.  The reference draws its batch indices with
``torch.randint(0, n, (B,), device=...)`` (/root/reference/algorithms/offline/iql.py:211-214),
whose stream differs between the CPU and CUDA generators and is therefore not a
cross-device contract; what IS the contract, and what this
file pins for the HIP kernels, is: uniform integers in [0, n) obtained as
``u32 % n`` (torch's ``random_from_to`` for ranges < 2**32), one independent draw
per batch row per step.

Stream definition shared with iqlpref_amd/csrc/philox.h:
  key     = (seed_lo, seed_hi)
  counter = (row_or_block, step_lo, step_hi, stream_id)
  stream 0 = replay indices (word 0 of the 4 outputs), stream 1/2 = dropout layer 1/2
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)

STREAM_INDEX = 0
STREAM_DROPOUT1 = 1
STREAM_DROPOUT2 = 2


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over the counter words (uint32 arrays); returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32).copy()
    c1 = np.broadcast_to(np.asarray(c1, dtype=np.uint32), c0.shape).copy()
    c2 = np.broadcast_to(np.asarray(c2, dtype=np.uint32), c0.shape).copy()
    c3 = np.broadcast_to(np.asarray(c3, dtype=np.uint32), c0.shape).copy()
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * M0
            p1 = c2.astype(np.uint64) * M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def sample_indices(seed, step, batch, n_rows):
    """Batch indices of optimisation step ``step`` (0-based): u32 % n_rows per row."""
    rows = np.arange(batch, dtype=np.uint32)
    r0, _, _, _ = philox4x32_10(rows, np.uint32(step & 0xFFFFFFFF),
                                np.uint32((step >> 32) & 0xFFFFFFFF),
                                np.uint32(STREAM_INDEX),
                                seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return (r0.astype(np.uint64) % np.uint64(n_rows)).astype(np.int64)


def dropout_stream(l):
    """Philox stream of the Dropout behind hidden layer ``l`` (0-based): 1 and 2 for the two hidden layers of
    the default networks, 8 + l for deeper ones (csrc/iql_deep.hip drop_stream; 3 is taken, see below)."""
    return STREAM_DROPOUT1 + l if l < 2 else 8 + l


def dropout_keep(seed, step, layer, batch, hidden, p):
    """Keep mask [batch, hidden] of the dropout layer with stream id ``layer`` (dropout_stream) at ``step``.

    One Philox call covers 4 consecutive batch rows of one hidden unit (the four
    accumulator registers a lane holds in the MFMA C/D layout): counter word 0 =
    (row // 4) * hidden + unit, output word (row % 4).  A unit is kept iff its
    word >= floor(p * 2**32)  (P(keep) = 1 - p).
    """
    assert batch % 4 == 0
    thr = np.uint32(min(int(p * 4294967296.0), 0xFFFFFFFF))
    blocks = np.arange((batch // 4) * hidden, dtype=np.uint32)
    r = philox4x32_10(blocks, np.uint32(step & 0xFFFFFFFF),
                      np.uint32((step >> 32) & 0xFFFFFFFF), np.uint32(layer),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(r, axis=0).reshape(4, batch // 4, hidden)  # [word][row//4][unit]
    words = words.transpose(1, 0, 2).reshape(batch, hidden)
    return words >= thr


STREAM_MLP_DROPOUT = 3


def mlp_dropout_keep(seed, call, layer, n_rows, width, p):
    """Keep mask [n_rows, width] of the Dropout behind hidden layer ``layer`` (0-based) of a
    stand-alone MLP forward in train mode (iqlhip_mlp_desc.dropout_*; k_mlp_f32): one Philox
    block per (row, 4 consecutive units), counter = (row, call, unit // 4 | layer << 16, stream 3),
    output word unit % 4; kept iff word >= max(1, floor(p * 2**32))."""
    thr = np.uint32(max(1, min(int(p * 4294967296.0), 0xFFFFFFFF)))
    nq = (width + 3) // 4
    rows = np.repeat(np.arange(n_rows, dtype=np.uint32), nq)
    cq = np.tile(np.arange(nq, dtype=np.uint32), n_rows) | np.uint32(layer << 16)
    r = philox4x32_10(rows, np.uint32(call & 0xFFFFFFFF), cq, np.uint32(STREAM_MLP_DROPOUT),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(r, axis=1).reshape(n_rows, nq * 4)[:, :width]
    return words >= thr
