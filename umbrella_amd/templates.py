"""Chat prompt scaffolds keyed like the reference's ``umbrella/templates.py`` (``Prompts`` / ``SysPrompts`` /
``ExtraPrompts``), so that its front-ends import unchanged.  The strings follow the public chat formats of the model
families (Llama-3 header tokens, ChatML for Qwen); the wording of the system prompts is this repository's own.
Not on the hot path."""

_L3_USER = "<|start_header_id|>user<|end_header_id|>\n\n{}<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n"

Prompts = {
    "meta-llama3": "\n" + _L3_USER,
    "llama3-code": _L3_USER,
    "qwen": "<|im_start|>user\n{}<|im_end|>\n<|im_start|>assistant\n",
    "gemma2": "{}",
    "gemma2-it": "<start_of_turn>user\n{}<end_of_turn>\n<start_of_turn>model\n",
    "mistral": "[INST] {} [/INST]",
}

SysPrompts = {
    "meta-llama3": "<|begin_of_text|><|start_header_id|>system<|end_header_id|>\n\n"
                   "You are a helpful, precise assistant.<|eot_id|>",
    "llama3-code": "<|begin_of_text|><|start_header_id|>system<|end_header_id|>\n\n"
                   "You are a careful programming assistant. Answer with working code.<|eot_id|>",
    "qwen": "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n",
    "gemma2": "",
    "gemma2-it": "",
    "mistral": "",
}

ExtraPrompts = {
    "llama3-code": "\nPrefer complete functions over fragments.",
}
