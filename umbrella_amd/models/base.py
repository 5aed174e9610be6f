"""Model-runtime face kept from the reference (umbrella/models/base.py:4-32)."""
from abc import ABC, abstractmethod


class LLMBase(ABC):
    def __init__(self) -> None:
        super().__init__()

    @abstractmethod
    def alloc(self, **kwargs):
        pass

    @abstractmethod
    def inference(self, input_ids, position_ids, attention_mask, storage_ids):
        pass

    @abstractmethod
    def graph_inference(self, input_ids, storage_ids, position_ids=None, attention_mask=None):
        pass
