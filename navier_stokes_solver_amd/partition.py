"""Host logic of the row partition: halo plans for the x-strip sharding.

The reference shards DoFs by mesh partition and lets Epetra build an ``Epetra_Import`` per
matrix.  Here every rank knows its ghost global ids (``LocalProblem.ghost_u/p``)
and all owned ranges; the plan says, per neighbour, which owned entries to send and which
ghost slots to fill — exactly the arrays ``nsk_set_halo_plan`` takes.
"""
from __future__ import annotations

import numpy as np


def owner_of(gids: np.ndarray, ranges: np.ndarray) -> np.ndarray:
    """Rank owning each global id, given the nranks+1 range offsets."""
    return np.searchsorted(np.asarray(ranges), np.asarray(gids), side="right") - 1


def build_halo_plan(rank: int, ranges: np.ndarray, ghosts_of_all_ranks: list) -> dict:
    """Plan of one space for ``rank``.

    ranges               : nranks+1 owned offsets (global ids)
    ghosts_of_all_ranks  : list over ranks of ascending ghost global-id arrays
    Returns peers / send_ptr / send_idx (owned local ids) / recv_ptr (ghost slots).
    """
    ranges = np.asarray(ranges, dtype=np.int64)
    nranks = len(ranges) - 1
    mine = np.asarray(ghosts_of_all_ranks[rank], dtype=np.int64)
    if len(mine) and np.any(np.diff(mine) <= 0):
        raise ValueError("ghost ids must be strictly ascending")
    my_owner = owner_of(mine, ranges) if len(mine) else np.zeros(0, dtype=np.int64)
    if np.any(my_owner == rank):
        raise ValueError("a ghost id lies in the owned range")
    begin, end = ranges[rank], ranges[rank + 1]
    send_lists = {}
    for q in range(nranks):
        if q == rank:
            continue
        g = np.asarray(ghosts_of_all_ranks[q], dtype=np.int64)
        sel = g[(g >= begin) & (g < end)]
        if len(sel):
            send_lists[q] = (sel - begin).astype(np.int32)
    recv_from = sorted(set(int(o) for o in my_owner))
    peers = sorted(set(recv_from) | set(send_lists))
    send_ptr, recv_ptr, send_idx = [0], [0], []
    pos = 0
    for q in peers:
        s = send_lists.get(q, np.zeros(0, np.int32))
        send_idx.append(s)
        send_ptr.append(send_ptr[-1] + len(s))
        cnt = int(np.count_nonzero(my_owner == q))
        # ascending ghost ids are grouped by owner because strips own contiguous ranges
        if cnt and not np.all(my_owner[pos:pos + cnt] == q):
            raise ValueError("ghost list is not grouped by owner")
        pos += cnt
        recv_ptr.append(recv_ptr[-1] + cnt)
    return dict(peers=np.asarray(peers, np.int32), send_ptr=np.asarray(send_ptr, np.int32),
                send_idx=np.concatenate(send_idx).astype(np.int32) if send_idx else np.zeros(0, np.int32),
                recv_ptr=np.asarray(recv_ptr, np.int32))
