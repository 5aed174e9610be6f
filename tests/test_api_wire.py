"""Wire API plumbing (CPU): framing round trip and server/client exchange with a stub engine."""
import socket
import threading
import time

from umbrella_amd.api.api_utils import receive_data, send_data
from umbrella_amd.api.client import APIClient
from umbrella_amd.api.server import APIServer


def test_framing_roundtrip():
    a, b = socket.socketpair()
    payload = {"context": "x" * 5000, "max_new_tokens": 7, "ids": list(range(300))}
    for wire in ("json", "pickle"):
        t = threading.Thread(target=send_data, args=(a, payload, wire))
        t.start()
        assert receive_data(b, allow_pickle=True) == payload
        t.join()
    # a pickle frame is refused unless the receiver opted in (unpickling is code execution)
    t = threading.Thread(target=send_data, args=(a, payload, "pickle"))
    t.start()
    try:
        receive_data(b)
        raise AssertionError("pickle frame accepted without allow_pickle")
    except ValueError:
        pass
    t.join()
    a.close(); b.close()


class _StubEngine:
    def __init__(self):
        self.calls = 0

    def generate(self, **kw):
        self.calls += 1
        kw.update(generated_text="ok", generated_tokens=[1, 2, 3], avg_accept_tokens=3.0, time_per_output_token=1.0)
        return kw


import pytest


@pytest.mark.parametrize("wire", ["json", "pickle"])
def test_server_client_exchange(wire):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    eng = _StubEngine()
    srv = APIServer(config={}, device="cpu", port=port, engine=eng, wire=wire)
    th = threading.Thread(target=srv.run, daemon=True)
    th.start()
    time.sleep(0.2)
    c = APIClient(port=port, retry_seconds=0.1, wire=wire)
    c.run()
    out = c.get_output(context="hello", max_new_tokens=4)
    assert out["processed"] is True and out["generated_tokens"] == [1, 2, 3] and out["context"] == "hello"
    out = c.get_output(input_ids=[5, 6], max_new_tokens=2)
    assert eng.calls == 2 and out["input_ids"] == [5, 6]
    c.close()
    srv.shutdown()
