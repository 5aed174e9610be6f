"""The `umbrella` alias package: every import the reference's front-ends perform (examples/*.py, app/*.py,
umbrella/api/*.py -- module path + names, recorded from the reference tree) resolves to the `umbrella_amd` objects."""
import importlib

import pytest

# (module, names) pairs of the reference's `from umbrella... import ...` statements
REFERENCE_IMPORTS = [
    ("umbrella.speculation.auto_engine", ["AutoEngine"]),
    ("umbrella.speculation.static_speculation_engine", ["StaticSpeculationEngine"]),
    ("umbrella.speculation.dynamic_speculation_engine", ["DynamicSpeculationEngine"]),
    ("umbrella.speculation.speculation_utils", ["make_causal_mask", "is_sentence_complete_regex", "find_first_element_position"]),
    ("umbrella.models.auto_model", ["AutoModelLM"]),
    ("umbrella.api.server", ["APIServer"]),
    ("umbrella.api.client", ["APIClient"]),
    ("umbrella.api.api_utils", ["send_data", "receive_data"]),
    ("umbrella.logging_config", ["setup_logger"]),
    ("umbrella.utils", ["TextColors"]),
    ("umbrella.templates", ["Prompts", "SysPrompts"]),
    ("umbrella.sequoia_utils", ["measure_acceptance_rate", "generate_sequoia_tree"]),
]


@pytest.mark.parametrize("module,names", REFERENCE_IMPORTS)
def test_reference_imports_resolve(module, names):
    mod = importlib.import_module(module)
    real = importlib.import_module(module.replace("umbrella", "umbrella_amd", 1))
    assert mod is real                                   # the same module object: no second copy of any state
    for n in names:
        assert getattr(mod, n) is getattr(real, n)


def test_alias_has_no_code_of_its_own_and_fails_cleanly():
    import umbrella
    assert umbrella.__path__ == []
    with pytest.raises(ModuleNotFoundError):
        importlib.import_module("umbrella.does_not_exist")
    from umbrella.templates import Prompts, SysPrompts
    for key in ("meta-llama3", "llama3-code", "qwen", "gemma2", "gemma2-it", "mistral"):
        assert key in Prompts and key in SysPrompts
    assert "{}" in Prompts["meta-llama3"]
