"""The engine contract of the reference (umbrella/speculation/base.py:4-59), enforced at class creation.

Instead of one ``@abstractmethod`` stub per method, ``BaseEngine`` lists the required names and refuses to be
instantiated by a subclass that leaves any of them out -- the same guarantee ``abc`` gives, with the method set in
one readable place."""

ENGINE_METHODS = (
    "initialize",                       # load models, build tables / graphs
    "prefill", "append",                # text in (tokenised by the engine)
    "_prefill", "_append",              # token ids in; return False instead of overflowing
    "build_tree", "verify",             # one speculation iteration = build_tree() then verify()
    "speculative_decoding",             # (dec_len, seconds, target_steps), streams text to stdout
    "generate", "generate_stream",      # API entry points (dict in / dict out, generator)
    "validate_status", "update_generation_args", "reset",
)


class BaseEngine:
    def __new__(cls, *args, **kwargs):
        missing = [m for m in ENGINE_METHODS if not callable(getattr(cls, m, None))]
        if missing:
            raise TypeError(f"Can't instantiate {cls.__name__}: missing engine methods {missing}")
        return super().__new__(cls)

    def __init__(self):
        super().__init__()
