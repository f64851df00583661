from .rag_e2e_base_model import AutoModelForRagE2E, Mode  # noqa: F401
from .retriever_only_base_model import AutoModelForSentenceEmbedding  # noqa: F401
