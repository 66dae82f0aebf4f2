from .backends.abstract import AttentionBackend, AttentionMetadata  # noqa: F401
from .layer import Attention  # noqa: F401
from .selector import get_attn_backend  # noqa: F401
