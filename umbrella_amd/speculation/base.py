"""Abstract engine surface, unchanged from the reference (umbrella/speculation/base.py:4-59)."""
from abc import ABC, abstractmethod


class BaseEngine(ABC):
    def __init__(self):
        super().__init__()

    @abstractmethod
    def initialize(self): ...

    @abstractmethod
    def verify(self): ...

    @abstractmethod
    def build_tree(self): ...

    @abstractmethod
    def prefill(self, text: str): ...

    @abstractmethod
    def append(self, text: str): ...

    @abstractmethod
    def _prefill(self, input_ids): ...

    @abstractmethod
    def _append(self, input_ids): ...

    @abstractmethod
    def speculative_decoding(self, max_new_tokens: int): ...

    @abstractmethod
    def validate_status(self): ...

    @abstractmethod
    def update_generation_args(self, **generation_args): ...

    @abstractmethod
    def reset(self): ...

    @abstractmethod
    def generate(self, **api_args): ...

    @abstractmethod
    def generate_stream(self, **api_args): ...
