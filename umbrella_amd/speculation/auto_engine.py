"""Engine factory.

Contract of the reference (umbrella/speculation/auto_engine.py:12-22) kept verbatim:
``AutoEngine.from_config(device, engine="static"|"dynamic", model=<target>, draft_model=<draft>, **engine_kwargs)``
returns a constructed (not yet initialised) engine; an unknown engine raises ``ValueError``, a missing model name
fails an ``assert``.  ``_ENGINE_MAPPING`` stays a plain dict so callers can register their own engines."""
from .dynamic_speculation_engine import DynamicSpeculationEngine as _Dynamic
from .static_speculation_engine import StaticSpeculationEngine as _Static


def _pop_required(config: dict, key: str):
    value = config.pop(key, None)
    assert value is not None, f"engine config lacks '{key}'"
    return value


class AutoEngine:
    _ENGINE_MAPPING = {"static": _Static, "dynamic": _Dynamic}

    @classmethod
    def from_config(cls, device: str, **config):
        kind = config.pop("engine", "dynamic")
        engine_cls = cls._ENGINE_MAPPING.get(kind)
        if engine_cls is None:
            raise ValueError(f"Engine type '{kind}' is not supported. Supported types: {sorted(cls._ENGINE_MAPPING)}")
        draft, target = _pop_required(config, "draft_model"), _pop_required(config, "model")
        return engine_cls(draft, target, device=device, **config)
