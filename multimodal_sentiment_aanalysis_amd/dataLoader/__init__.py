from .DataLoader import DevicePrefetcher, MultimodalDataLoader, SyntheticPairs  # noqa: F401
from .MultiTaskTrainer import MultiTaskTrainer  # noqa: F401
