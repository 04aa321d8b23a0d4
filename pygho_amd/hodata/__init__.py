"""Data-format side of the hot path: the padded-batch builders of the dense (MaskedTensor) layout
(reference pygho/hodata/MaData.py).  The sparse layout's collate lives in ``pygho_amd.collate``."""
from .MaData import batch2dense, to_dense_adj, to_dense_tuplefeat, to_dense_x, to_sparse_adj  # noqa: F401
