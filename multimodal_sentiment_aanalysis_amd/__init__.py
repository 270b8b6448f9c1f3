"""MI355X-native text+image sentiment hot path. Public surface mirrors the reference's modules."""
from ._lib import MmsaError, load  # noqa: F401
from .engine import BERT_BASE, BERT_LARGE, RESNET50, RESNET101, CrossEntropyLoss, materialize  # noqa: F401
from .MultimodalModel import (Classifier, CrossModalTransformer, MultiModalEncoder, MultimodalTransformerModel,  # noqa: F401
                              ProjectionHead)
