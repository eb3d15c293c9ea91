"""Doc-range sharding of one table across ranks (one process per GPU) and the per-batch top-k exchange.

Every rank owns a contiguous doc-id range as a complete small index: postings never cross shards, so
set algebra needs no communication. What must be global is what BM25 reads: N, avgdl and each gram's document
frequency (so idf — and therefore every score — is identical on all ranks); they are summed once at load time. Per
batch the ranks exchange only their per-query top-(offset+limit): ONE all-gather of a packed blob (64-bit keys and
totals, then 32-bit doc ids and counts, see mgx_batch_export_topk) followed by a device-side merge with the ResultSorter::SortByScore comparator. RCCL has no
custom reduction operator, so the north star's "all-reduce of top-k" is realised as all-gather + local merge.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import engine


def shard_range(n_docs_total, rank, world):
    """Contiguous doc range of `rank`: (number of docs before it, its doc count). Doc ids are 1-based."""
    per = (n_docs_total + world - 1) // world
    first = min(rank * per, n_docs_total)
    return first, min(per, n_docs_total - first)


def _device_for_backend():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def allreduce_sum_i64(values):
    """Sum of an int64 vector over all ranks (identity when not distributed)."""
    a = np.ascontiguousarray(values, dtype=np.int64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return a
    t = torch.from_numpy(a.copy()).to(_device_for_backend())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def global_table_stats(local_keys, local_sizes, local_doc_count, local_total_len):
    """-> (global posting size per LOCAL gram id, global BM25 doc_count, global total_len, {gram: global size} over
    the union of all shards' dictionaries).

    Shards may hold different gram dictionaries; sizes are aligned by gram bytes."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1:
        sizes = np.asarray(local_sizes, dtype=np.int64)
        return sizes, int(local_doc_count), int(local_total_len), dict(zip(local_keys, sizes.tolist()))
    gathered = [None] * world
    dist.all_gather_object(gathered, list(local_keys))
    union = sorted(set(k for ks in gathered for k in ks))
    pos = {k: i for i, k in enumerate(union)}
    vec = np.zeros(len(union) + 2, dtype=np.int64)
    for k, s in zip(local_keys, local_sizes):
        vec[pos[k]] = s
    vec[-2], vec[-1] = local_doc_count, local_total_len
    tot = allreduce_sum_i64(vec)
    sizes = np.asarray([tot[pos[k]] for k in local_keys], dtype=np.int64)
    return sizes, int(tot[-2]), int(tot[-1]), {k: int(tot[i]) for k, i in pos.items()}


class Comm:
    """mgx_comm (include/mygram_gpu.h): this rank's RCCL communicator over the ranks of a sharded table. The 128-byte id
    is drawn on rank 0 and handed round with torch.distributed (whatever backend the job runs: plumbing only — the
    per-batch collective itself is RCCL inside the library, search_pipeline::BatchExecutor::Options::comm)."""

    def __init__(self, rank=None, world=None, device=None):
        import ctypes as C
        from . import _capi
        L = _capi.load()
        init = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank() if init else 0) if rank is None else rank
        self.world = (dist.get_world_size() if init else 1) if world is None else world
        self.device = torch.cuda.current_device() if device is None else device
        ident = [None]
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            _capi.check(L.mgx_comm_unique_id(buf))
            ident[0] = bytes(buf)
        if self.world > 1:
            dist.broadcast_object_list(ident, src=0)
        h = C.c_void_p()
        raw = (C.c_uint8 * 128).from_buffer_copy(ident[0])
        _capi.check(L.mgx_comm_create(raw, self.rank, self.world, self.device, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            from . import _capi
            _capi.load().mgx_comm_destroy(self._h)
            self._h = None


def _device_bytes(ptr, nbytes):
    """uint8 torch view of `nbytes` of library-owned device memory at `ptr` (current device), without a copy."""
    class _Arr:
        __cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Arr(), device=torch.device("cuda", torch.cuda.current_device()))


class ShardedTable:
    """One rank's shard of a table plus the global statistics every rank agrees on."""

    def __init__(self, corpus, first_doc_id, device=0, ngram_size=2, kanji_ngram_size=0, cross_boundary=True,
                 dense_threshold=0.0, n_threads=0):
        self.rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.device = device
        self.index = engine.Index(corpus=corpus, first_doc_id=first_doc_id, ngram_size=ngram_size,
                                  kanji_ngram_size=kanji_ngram_size, cross_boundary=cross_boundary, device=device,
                                  dense_threshold=dense_threshold, n_threads=n_threads)
        c = self.index.columns
        keys = [c.gram(g) for g in range(c.n_grams)]
        sizes = np.diff(c.offsets.astype(np.int64))
        gsizes, n, total_len, gdict = global_table_stats(keys, sizes, c.bm25_doc_count, c.bm25_total_len)
        self.index._global_sizes = gsizes
        # a gram another shard holds but this one lacks is an empty operand here, not an unknown term: every rank then
        # compiles the same batch layout (same queries on the device, same text-level terms in the df buffer)
        self.index._global_dict = gdict
        self.index.total_docs = n
        self.index.avg_doc_length = (total_len / n) if n else 0.0
        self.keys, self.global_sizes = keys, gsizes
        # rehearsal switch: run the export / all-gather / merge path even with a single rank
        self.force_exchange = bool(dist.is_available() and dist.is_initialized() and
                                   __import__("os").environ.get("MGX_FORCE_EXCHANGE"))

    def prepare(self, queries):
        return self.index.prepare(queries)

    def _buffers(self, batch):
        """Per batch: this rank's packed blob and the gathered blobs of every rank. One rank's blob is
        [keys | totals] as u64 followed by [docs | counts] as u32 (mgx_batch_export_topk), padded to a multiple of 8
        bytes, so ONE all-gather moves both. SORT _score batches keep their result in that layout inside the library
        (mgx_batch_export_buffer): the all-gather reads it in place."""
        # The buffers live ON the batch object: the zero-copy views alias memory the batch owns (d_export), so they must
        # die with it. (Keying a table-wide cache on id(batch) handed a recycled id the previous batch's freed pointers.)
        if getattr(batch, "_exchange", None) is None:
            dev = torch.device("cuda", self.index.device_index.device)
            ptr, nbytes, off32 = batch.export_buffer()
            if ptr:
                class _Arr:  # __cuda_array_interface__ view of library-owned memory
                    __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

                mine, copy = torch.as_tensor(_Arr(), device=dev), False
            else:
                stride = batch.topk_stride()
                n = batch.n * stride + batch.n          # elements of each blob
                off32, nbytes = 8 * n, (12 * n + 7) // 8 * 8
                mine, copy = torch.empty(nbytes, dtype=torch.uint8, device=dev), True
            batch._exchange = (off32, nbytes, mine, copy,
                               torch.empty(nbytes * self.world, dtype=torch.uint8, device=dev))
        return batch._exchange

    def _df_tensor(self, batch, ptr, n):
        """The batch's device df array (u64; counts stay far below 2^63) as an int64 torch tensor, without a copy."""
        if getattr(batch, "_df_view", None) is None:
            dev = self.index.device_index.device

            class _Arr:  # __cuda_array_interface__ view of library-owned memory
                __cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}

            batch._df_view = torch.as_tensor(_Arr(), device=torch.device("cuda", dev))
        return batch._df_view

    def run(self, batch):
        """Executes `batch` on this shard, exchanges per-shard top-k and merges: afterwards batch.fetch*() returns
        the table-wide page and total on every rank."""
        stream = torch.cuda.current_stream().cuda_stream
        exchange = self.world > 1 or self.force_exchange
        if exchange and dist.get_backend() == "nccl":
            # the production path: both collectives are RCCL calls inside the library (mgx_batch_exchange[_df])
            if getattr(self, "comm", None) is None:
                self.comm = Comm(device=self.device)  # (the table's device, not whatever torch's current one is)
            if batch.n:
                batch.exchange_df(self.comm, stream)
            batch.execute_sharded(self.comm, stream)
            if batch.n:
                batch.exchange(self.comm, stream)
            return
        if exchange and batch.n:
            ptr, n_tt = batch.df_buffer()
            if n_tt:
                # text-level terms: df is the table-wide count (PopulateTermDocumentFrequency over every shard's
                # candidates), so the per-shard counts are summed before any rank takes idf
                batch.count_df(stream)
                df = self._df_tensor(batch, ptr, n_tt)
                if dist.get_backend() == "nccl":
                    dist.all_reduce(df, op=dist.ReduceOp.SUM)
                else:
                    h = df.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    df.copy_(h)
        if exchange and batch.n:
            # the seed keys' all-gather of mgx_batch_execute_sharded, through torch.distributed (gloo stages through host
            # memory: the CPU rehearsal; the device buffers are the batch's own)
            def gather(mine_ptr, all_ptr, nbytes):
                torch.cuda.current_stream().synchronize()
                mine_t = _device_bytes(mine_ptr, nbytes)
                all_t = _device_bytes(all_ptr, nbytes * self.world)
                if dist.get_backend() == "nccl":
                    dist.all_gather_into_tensor(all_t, mine_t)
                else:
                    parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
                    dist.all_gather(parts, mine_t.cpu())
                    all_t.copy_(torch.cat(parts))
                torch.cuda.current_stream().synchronize()
            batch.execute_gather(self.world, gather, stream)
        else:
            batch.execute(stream)
        if not exchange:
            return
        off32, nbytes, mine, copy, gathered = self._buffers(batch)
        if copy:
            batch.export_topk(mine.data_ptr(), mine.data_ptr() + off32, stream)
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(gathered, mine)
        else:  # gloo: stage through host memory (CPU rehearsal of the exchange; RCCL is the production path)
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, mine.cpu())
            gathered.copy_(torch.cat(parts))
        batch.merge_shards(self.world, gathered.data_ptr(), gathered.data_ptr() + off32, stream,
                           pitch64=nbytes // 8, pitch32=nbytes // 4)
