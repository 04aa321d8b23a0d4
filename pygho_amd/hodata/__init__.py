"""Data-format side of the hot path: the padded-batch builders of the dense (MaskedTensor) layout (reference
pygho/hodata/MaData.py) and the sparse-layout helpers (pygho/hodata/SpData.py: key parsing, per-graph precomputation of the
message triples, batch -> SparseTensor) and the tuple samplers (pygho/hodata/SpTupleSampler.py) as device kernels.  Device
collation of the sparse layout lives in ``pygho_amd.collate``."""
from .MaData import batch2dense, to_dense_adj, to_dense_tuplefeat, to_dense_x, to_sparse_adj  # noqa: F401
from .SpData import batch2sparse, parsekey, parseop, sp_datapreprocess  # noqa: F401
from .SpTupleSampler import I2Sampler, KhopSampler, i2_sample, khop_sample  # noqa: F401
