"""Front-ends of SURVEY 8(f)4 (the reference's app/chatbot.py, app/api.py, app/gradio_chat.py): the turn logic against a
scripted engine on CPU, and the terminal chat + the streaming reply on a real tiny HIP engine (GPU)."""
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class ScriptedEngine:
    """records the calls a front-end makes; `room` = how many prompts still fit"""

    def __init__(self, room=99, stream=("Hel", "Hello", "Hello there")):
        self.calls, self.room, self.stream, self.kw = [], room, stream, None

    def prefill(self, text):
        self.calls.append(("prefill", text)); self.room -= 1
        return self.room >= 0

    def append(self, text):
        self.calls.append(("append", text)); self.room -= 1
        return self.room >= 0

    def speculative_decoding(self, max_new_tokens=128):
        self.calls.append(("decode", max_new_tokens))
        return (17, 0.5, 6)

    def validate_status(self):
        return self.room > 0

    def generate_stream(self, **kw):
        self.kw = kw
        for i, t in enumerate(self.stream):
            yield t, f"perf {i}"


def test_chat_session_turns_follow_the_reference_loop():
    """app/chatbot.py:27-51: system prompt only on the first turn (prefill), later turns through append, BYE ends, a context
    that no longer fits ends, max_turns bounds the loop."""
    from app.chat import ChatSession, TurnResult
    e = ScriptedEngine()
    s = ChatSession(e, "<sys>", "[u]{}[/u]", generation_length=64, max_turns=3)
    assert s.say("hi") is TurnResult.ANSWERED and s.say("more") is TurnResult.ANSWERED
    assert e.calls == [("prefill", "<sys>[u]hi[/u]"), ("decode", 64), ("append", "[u]more[/u]"), ("decode", 64)]
    assert s.say("BYE") is TurnResult.GOODBYE and len(e.calls) == 4                      # nothing reaches the engine
    assert [t[0] for t in s.transcript] == ["hi", "more"] and s.transcript[0][1] == (17, 0.5, 6)
    # the loop: scripted terminal; ends on max_turns
    e2 = ScriptedEngine()
    typed = iter(["a", "b", "c", "d"])
    out = []
    s2 = ChatSession(e2, "S", "{}", max_turns=3)
    assert s2.run(read=lambda prompt: next(typed), tell=out.append) is TurnResult.ANSWERED and s2.turns_done == 3
    # overflow on append: engine returns False -> the session stops without decoding
    e3 = ScriptedEngine(room=1)
    s3 = ChatSession(e3, "S", "{}")
    assert s3.say("x") is TurnResult.OUT_OF_CONTEXT            # answered, but nothing fits afterwards (validate_status False)
    e4 = ScriptedEngine(room=0)
    assert ChatSession(e4, "S", "{}").say("x") is TurnResult.OUT_OF_CONTEXT and ("decode", 256) not in e4.calls


def test_stream_reply_reframes_the_history_and_keeps_state():
    """app/gradio_chat.py:26-58: every message re-frames system prompt + all earlier exchanges + the new input, yields the growing
    answer with the performance line, appends the finished exchange to the state."""
    from app.gradio_chat import frame_conversation, stream_reply
    hist = [("q1", "a1")]
    assert frame_conversation("S|", "<{}>", hist, "q2") == "S|<q1>a1<q2>"
    e = ScriptedEngine()
    frames = list(stream_reply(e, "S|", "<{}>", hist, "q2", max_new_tokens=40, temperature=0.3, top_p=0.8, repetition_penalty=1.1))
    assert e.kw == {"context": "S|<q1>a1<q2>", "max_new_tokens": 40, "temperature": 0.3, "topp": 0.8, "repetition_penalty": 1.1}
    assert [f[2] for f in frames] == ["perf 0", "perf 1", "perf 2"] and all(f[3] == "" for f in frames)
    assert frames[-1][0] == [("q1", "a1"), ("q2", "Hello there")] and frames[0][0][-1] == ("q2", "Hel")
    assert hist == [("q1", "a1"), ("q2", "Hello there")]
    with pytest.raises(RuntimeError, match="gradio"):
        from app.gradio_chat import build_ui
        build_ui(e, "m", "S", "{}")


def test_api_demo_requests_and_wire_round_trip():
    """app/api.py:22-66: the demo's requests (text, and ids when a tokenizer exists) against a served engine over TCP."""
    import socket
    import threading
    from app.api import requests_for
    from umbrella_amd.api.client import APIClient
    from umbrella_amd.api.server import APIServer

    class Tok:
        def encode(self, t):
            return [len(t) % 97, 5, 6]
    reqs = requests_for({"template": "qwen"}, questions=("why?",), tokenizer=Tok(), max_new_tokens=8)
    assert [sorted(r) for r in reqs] == [["context", "max_new_tokens", "temperature"], ["input_ids", "max_new_tokens", "temperature"]]
    assert reqs[0]["context"].endswith("<|im_start|>assistant\n") and "why?" in reqs[0]["context"]

    class Echo:
        def generate(self, **kw):
            kw["generated_text"] = "ok:" + str(kw.get("context", kw.get("input_ids")))[:20]
            kw.update(generated_tokens=[1, 2], avg_accept_tokens=2.0, time_per_output_token=1.0)
            return kw
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    srv = APIServer(config={}, port=port, max_client=1, engine=Echo())
    th = threading.Thread(target=srv.run, daemon=True)
    th.start()
    c = APIClient(port=port)
    c.run()
    try:
        outs = [c.get_output(**r) for r in reqs]
    finally:
        c.close()
        srv.close() if hasattr(srv, "close") else None
    assert outs[0]["generated_text"].startswith("ok:") and outs[1]["generated_tokens"] == [1, 2]


@pytest.mark.gpu
def test_terminal_chat_and_stream_on_a_tiny_hip_engine(capsys):
    """the two front-ends over a REAL engine (tiny target + self-draft, static 3x4, id tokenizer): two chat turns (prefill, append)
    print decoded ids, the second conversation streams cumulative text whose last frame is the whole answer."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import __graft_entry__ as ge
    ge.build()
    from helpers import load_golden
    from hip_helpers import static_engine
    from app.chat import ChatSession, TurnResult
    from app.gradio_chat import stream_reply
    g = load_golden()
    eng, _ = static_engine(g, torch.device("cuda:0"), torch.float16, self_draft=True)
    s = ChatSession(eng, "11 250 7 ", "{} 42 ", generation_length=16, max_turns=4)
    assert s.say("1999 9001 345") is TurnResult.ANSWERED
    n1 = eng.num_nodes
    assert s.say("77 5") is TurnResult.ANSWERED and eng.num_nodes > n1
    assert s.transcript[0][1][0] >= 16 and capsys.readouterr().out.strip()          # tokens were emitted and printed
    eng.reset()
    hist = []
    frames = list(stream_reply(eng, "11 250 7 ", "{} 42 ", hist, "1999 9001", max_new_tokens=16, temperature=0.0))
    assert len(frames) >= 2 and frames[-1][0][-1][0] == "1999 9001" and len(hist) == 1
    texts = [f[0][-1][1] for f in frames]
    assert all(texts[i + 1].startswith(texts[i][:len(texts[i]) // 2]) for i in range(len(texts) - 1)) and hist[0][1] == texts[-1]
    assert "Avg Accept Tokens" in frames[-1][2]
