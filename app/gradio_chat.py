"""Streaming web chat (the reference's ``app/gradio_chat.py:26-123``): the whole conversation is re-framed into one prompt
per message and streamed through ``engine.generate_stream``; sliders for max_new_tokens / temperature / top_p /
repetition_penalty, a performance line, a clear button.

The conversation logic (`frame_conversation`, `stream_reply`) is plain Python and tested on its own; `build_ui` needs
the optional ``gradio`` package and says so when it is missing (it is not part of this image).

    python -m app.gradio_chat --configuration configs/static_70b_awq_on_device.yaml
"""
from __future__ import annotations

import argparse
from typing import Iterator


def frame_conversation(system_prompt: str, user_template: str, history, user_input: str) -> str:
    """system prompt, then every earlier (user, assistant) exchange re-framed, then the new user message."""
    parts = [system_prompt]
    for said, answered in history:
        parts.append(user_template.format(said))
        parts.append(answered)
    parts.append(user_template.format(user_input))
    return "".join(parts)


def stream_reply(engine, system_prompt: str, user_template: str, history: list, user_input: str, max_new_tokens: int = 128,
                 temperature: float = 0.6, top_p: float = 0.9, repetition_penalty: float = 1.05) -> Iterator[tuple]:
    """Yield (chat pairs to display, state, performance line, cleared input box) while the answer grows; when the stream
    ends the finished exchange is appended to `history` (the state the UI keeps between messages)."""
    prompt = frame_conversation(system_prompt, user_template, history, user_input)
    answer = ""
    # the engine's knob is spelled `topp` (update_generation_args); the reference's UI passes `top_p`, which its engine ignores
    for answer, perf in engine.generate_stream(context=prompt, max_new_tokens=int(max_new_tokens), temperature=float(temperature),
                                               topp=float(top_p), repetition_penalty=float(repetition_penalty)):
        shown = list(history) + [(user_input, answer)]
        yield shown, shown, perf, ""
    history.append((user_input, answer))


def build_ui(engine, model_name: str, system_prompt: str, user_template: str):
    try:
        import gradio as gr
    except ImportError as e:                                    # pragma: no cover - optional dependency
        raise RuntimeError("app.gradio_chat needs the optional `gradio` package (pip install gradio); "
                           "app.chat is the terminal front-end") from e
    with gr.Blocks(title="umbrella_amd chat") as demo:
        with gr.Row():
            gr.Textbox(value=model_name, label="Model", interactive=False)
            perf_box = gr.Textbox(label="Performance")
        chat = gr.Chatbot(label="Conversation")
        box = gr.Textbox(label="Input", placeholder="Type here ...")
        with gr.Row():
            n_new = gr.Slider(32, 512, value=128, step=1, label="max_new_tokens")
            temp = gr.Slider(0.0, 1.0, value=0.6, step=0.05, label="temperature")
        with gr.Row():
            top_p = gr.Slider(0.0, 1.0, value=0.9, step=0.05, label="top_p")
            rep = gr.Slider(1.0, 2.0, value=1.05, step=0.05, label="repetition_penalty")
        clear = gr.Button("clear")
        state = gr.State([])

        def on_message(text, history, a, b, c, d):
            yield from stream_reply(engine, system_prompt, user_template, history, text, a, b, c, d)

        box.submit(on_message, [box, state, n_new, temp, top_p, rep], [chat, state, perf_box, box])
        clear.click(lambda: ([], "", []), None, [chat, perf_box, state], queue=False)
    return demo


def main(argv=None):
    from umbrella_amd.speculation.auto_engine import AutoEngine
    from umbrella_amd.templates import Prompts, SysPrompts
    from umbrella_amd.utils import load_config
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--configuration", default="configs/static_70b_awq_on_device.yaml")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--share", action="store_true")
    args = ap.parse_args(argv)
    config = dict(load_config(args.configuration))
    template = config.pop("template", "meta-llama3")
    for k in ("generation_length", "max_turns"):
        config.pop(k, None)
    engine = AutoEngine.from_config(args.device, **config)
    engine.initialize()
    build_ui(engine, config.get("model", ""), SysPrompts[template], Prompts[template]).launch(share=args.share)


if __name__ == "__main__":
    main()
