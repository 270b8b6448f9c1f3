from .DataLoader import MultimodalDataLoader, SyntheticPairs  # noqa: F401
