"""Corpus-sharded dense retrieval with ONE collective (BASELINE.json configs[4]: mMARCO-fr, 8.8 M passages).

Reference spec: InformationRetrievalEvaluatorCustom.compute_metrices (src/utils/sentence_transformers.py:314-393):
corpus in chunks, per chunk score -> torch.topk -> heap merge keeping max_k = 1000 per query.  The reference does
this on one GPU, one query at a time (1.2 M launches for mMARCO).  Here:

  * the corpus-embedding matrix is row-sharded over the ranks (one process per GPU, 288 GB HBM each);
  * every rank scores ALL queries against its shard in document chunks: fp32-MFMA GEMM -> per-row top-k
    (csrc/sort.hip) -> merge into the running top-k -- no host round trip;
  * ONE all-gather of the per-shard [Q, k] (score fp32, id int64) lists over RCCL/xGMI (8.2 MB + 8.2 MB per rank at
    Q = 1024, k = 1000: < 1 % of the GEMM time) and an identical local G-way merge on every rank;
  * the query ENCODER is data-parallel over the queries: every rank runs the transformer on its 1/G of the batch and the
    [Q, 768] embeddings (3 MB) are all-gathered -- the only other exchange step.
Ties are broken by ascending global document id everywhere, so the result does not depend on the number of shards.
"""
from __future__ import annotations

import torch


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced row shards: the first n % world shards get one extra row."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allgather_rows(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """Rows sharded by shard_bounds(total, world, rank) -> the full [total, d] tensor on every rank, in rank order:
    one all-gather of equal-size (zero-padded) blocks."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    per = -(-total // world)
    block = local.new_zeros((per, local.shape[1]))
    block[: local.shape[0]] = local
    out = local.new_empty((world * per, local.shape[1]))
    dist.all_gather_into_tensor(out, block, group=group)
    if total == world * per:
        return out
    return torch.cat([out[r * per: r * per + (hi - lo)] for r in range(world) for lo, hi in [shard_bounds(total, world, r)]])


def allgather_topk(local_scores: torch.Tensor, local_ids: torch.Tensor, group=None, merge_fn=None):
    """local [Q, k] lists (each sorted by score desc, id asc; padding = (-inf, -1)) -> global [Q, k] on every rank.
    merge_fn([G,Q,k] scores, [G,Q,k] ids) -> ([Q,k], [Q,k]); defaults to the HIP merge."""
    import torch.distributed as dist
    if merge_fn is None:
        from . import ops
        merge_fn = ops.topk_merge
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return merge_fn(local_scores.unsqueeze(0).contiguous(), local_ids.unsqueeze(0).contiguous())
    Q, k = local_scores.shape
    gs = torch.empty((world, Q, k), dtype=local_scores.dtype, device=local_scores.device)
    gi = torch.empty((world, Q, k), dtype=local_ids.dtype, device=local_ids.device)
    # concatenated-along-dim-0 output form: accepted by RCCL and gloo alike
    dist.all_gather_into_tensor(gs.view(world * Q, k), local_scores.contiguous(), group=group)
    dist.all_gather_into_tensor(gi.view(world * Q, k), local_ids.contiguous(), group=group)
    return merge_fn(gs, gi)


class ShardedDenseIndex:
    """One rank's shard of the L2-normalised corpus embeddings + the chunked score -> top-k loop."""

    CHUNK = 8 * 28672   # documents per GEMM launch: 8 sort-kernel rows per query
    CAP = 7168          # candidate slots per row and chunk on the streaming path (k + CAP = one 8192-key sort row at k = 1024)
    FUSED = True        # after the head, score and filter in one kernel (fz_dot_scores_filter_f32); False: GEMM, then the filter pass
    HEAD = 28672        # at most this many leading documents get the exact top-k (one sort-kernel row); 8 k of them (>= 8192) are enough

    def __init__(self, Dn_local: torch.Tensor, id_base: int, group=None):
        self.Dn, self.id_base, self.group = Dn_local, int(id_base), group
        self.last_overflow = 0   # windows of the last local_topk whose candidate buffers overflowed (each was redone exactly, ops.TopkStream)

    def local_topk(self, Qn: torch.Tensor, k: int, streaming: bool = True, mark=None):
        """Chunked score -> top-k over this shard.  The first HEAD documents get an exact top-k (one sort-kernel row per query);
        after that only scores above a query's running k-th best can enter, so every later chunk goes through the streaming
        threshold filter and the candidates are folded into the list a few times per shard (ops.TopkStream).  A window in which a
        row overflowed its candidate buffer is redone exactly, on its own (flag read at every fold: 4 small reads per shard).
        `mark(name)`: optional instrumentation hook (bench.py records a HIP event per call).
        Everything runs on ONE stream: issuing the top-k work of chunk c on a second stream under the GEMM of chunk c + 1 was
        measured twice and lost both times (17.3 vs 15.5 ms per 1.1 M-document shard in round 2: the persistent GEMM owns every
        CU, and what squeezes in next to it costs the matrix pipe more than it hides)."""
        from . import ops
        mark = mark or (lambda name: None)
        n = self.Dn.shape[0]
        head = min(self.HEAD, max(8192, -(-8 * k // 4096) * 4096))   # 8192 at k = 1000: a 0.09 ms sort instead of 0.35, one fold more
        streaming = streaming and k + self.CAP <= 35840 and k <= head // 8 and n > head
        best_s = best_i = stream = None
        if streaming and self.FUSED and Qn.shape[1] % 4 == 0:
            # the head's scores are materialised and ranked exactly; everything after it goes through the GEMM whose epilogue is the
            # threshold filter: no score plane, no filter pass -- per shard 4.5 GB less written and 4.5 GB less read
            S = ops.dot_scores(Qn, self.Dn[:head]); mark("shard_gemm")
            bs, bi = ops.topk_rows(S, k, id_base=self.id_base)
            stream = ops.TopkStream(bs, bi, seen=head, cap=self.CAP); mark("shard_topk_stream")
            for c0 in range(head, n, self.CHUNK):
                c1 = min(n, c0 + self.CHUNK)
                stream.feed_gemm(Qn, self.Dn[c0:c1], self.id_base + c0, mark=mark)
            best_s, best_i, _ = stream.result(); mark("shard_topk_stream")
            self.last_overflow = stream.windows_redone
            return best_s, best_i
        for c0 in range(0, max(n, 1), self.CHUNK):
            c1 = min(n, c0 + self.CHUNK)
            S = ops.dot_scores(Qn, self.Dn[c0:c1]); mark("shard_gemm")
            if streaming:
                lo = 0
                if stream is None:
                    lo = min(head, c1 - c0)
                    bs, bi = ops.topk_rows(S[:, :lo], k, id_base=self.id_base + c0)
                    stream = ops.TopkStream(bs, bi, seen=lo, cap=self.CAP)
                stream.feed(S[:, lo:], self.id_base + c0 + lo); mark("shard_topk_stream")
                if stream.unrepairable:   # a window of scores the stream does not hold overflowed: the rest of the streaming pass would be wasted
                    break
            elif best_s is None:
                best_s, best_i = ops.topk_rows(S, k, id_base=self.id_base + c0); mark("shard_topk_exact")
            else:   # exact path: per-chunk top-k, then merge two id-ascending lists (chunks arrive in id order)
                s, i = ops.topk_rows(S, k, id_base=self.id_base + c0)
                best_s, best_i = ops.topk_merge(torch.stack([best_s, s]), torch.stack([best_i, i])); mark("shard_topk_exact")
        self.last_overflow = 0
        if stream is not None:
            best_s, best_i, flag = stream.result(); mark("shard_topk_stream")
            if int(flag.item()) != 0:   # a window of materialised scores overflowed (the stream holds no score plane: one chunk alive at a
                res = self.local_topk(Qn, k, streaming=False, mark=mark)   # time): this shard again on the exact path
                self.last_overflow = 1
                return res
        return best_s, best_i

    def search(self, Qn: torch.Tensor, k: int = 1000, mark=None):
        s, i = self.local_topk(Qn, k, mark=mark)
        out = allgather_topk(s, i, self.group)
        if mark: mark("allgather_merge")
        return out
