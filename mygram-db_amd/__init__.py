"""mygram-db_amd — MI355X-native query hot path of MygramDB (posting-list set algebra + BM25 + top-k).

The product is libmygram_gpu.so (HIP kernels for gfx950 behind the C ABI of include/mygram_gpu.h). This package is the
thin Python mirror of the reference's operator interface over that ABI, used by the tests, bench.py and the multi-GPU
driver. Import name: `mygram_db_amd` (the directory name has a hyphen; see load_package() in __graft_entry__.py).
"""
from . import _capi, engine  # noqa: F401
from .engine import (Columns, Corpus, DeviceIndex, Index, PreparedBatch, compute_idf,  # noqa: F401
                     generate_query_ngrams)
