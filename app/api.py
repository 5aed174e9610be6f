"""Wire-API demo: ``--server`` serves one engine on a TCP port; without it, a client sends two requests -- once as text
(``context``), once as token ids (``input_ids``) -- and prints the answers.  Counterpart of the reference's
``app/api.py:18-77`` over ``umbrella_amd.api`` (4-byte length framing; JSON payloads by default, ``--wire pickle`` to talk
to an unmodified reference peer on a trusted host).

    python -m app.api --server --configuration configs/static_70b_awq_on_device.yaml
    python -m app.api          --configuration configs/static_70b_awq_on_device.yaml
"""
from __future__ import annotations

import argparse

QUESTIONS = ("Summarise what speculative decoding is in 100 words.",
             "Explain locality-sensitive hashing in 100 words.")


def requests_for(config: dict, questions=QUESTIONS, tokenizer=None, max_new_tokens: int = 512):
    """The demo's request list: every question once as templated text and -- when a tokenizer is at hand -- once as ids."""
    from umbrella_amd.templates import Prompts, SysPrompts
    template = config.get("template", "meta-llama3")
    texts = [SysPrompts[template] + Prompts[template].format(q) for q in questions]
    reqs = [{"context": t, "max_new_tokens": max_new_tokens, "temperature": 0.0} for t in texts]
    if tokenizer is not None:
        reqs += [{"input_ids": list(tokenizer.encode(t)), "max_new_tokens": max_new_tokens, "temperature": 0.0} for t in texts]
    return reqs


def serve(config: dict, port: int, max_client: int, wire: str):
    from umbrella_amd.api.server import APIServer
    APIServer(config=config, port=port, max_client=max_client, wire=wire).run()


def ask(config: dict, port: int, wire: str, tell=print):
    from umbrella_amd.api.client import APIClient
    from umbrella_amd.utils import TextColors
    tok = None
    try:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(config.get("model"))
    except Exception:                                            # no tokenizer files offline: the text requests still run
        pass
    client = APIClient(port=port, wire=wire)
    client.run()
    answers = []
    try:
        for req in requests_for(config, tokenizer=tok):
            out = client.get_output(**req)
            answers.append(out)
            tell(TextColors.colorize(out["generated_text"], "cyan"))
    finally:
        client.close()
    return answers


def main(argv=None):
    from umbrella_amd.utils import load_config
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--configuration", default="configs/static_70b_awq_on_device.yaml")
    ap.add_argument("--port", type=int, default=65432)
    ap.add_argument("--max_client", type=int, default=1)
    ap.add_argument("--wire", default="json", choices=["json", "pickle"])
    ap.add_argument("--server", action="store_true", help="serve; otherwise act as the demo client")
    args = ap.parse_args(argv)
    config = load_config(args.configuration)
    if args.server:
        serve(config, args.port, args.max_client, args.wire)
    else:
        ask(config, args.port, args.wire)


if __name__ == "__main__":
    main()
