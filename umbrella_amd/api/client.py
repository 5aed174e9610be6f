"""Blocking client of the wire API.

Same surface as the reference client (umbrella/api/client.py): ``APIClient(port, host).run()``, then
``get_output(**generate_kwargs) -> dict`` per request and ``close()``; also usable as a context manager."""
import socket
import time

from ..logging_config import setup_logger
from ..utils import TextColors
from . import api_utils

_log = setup_logger()


def _dial(host: str, port: int, pause: float) -> socket.socket:
    """Keep trying until the server accepts (it may still be loading a model)."""
    attempt = 0
    while True:
        attempt += 1
        try:
            return socket.create_connection((host, port))
        except ConnectionRefusedError:
            _log.info(TextColors.colorize(f"attempt {attempt}: {host}:{port} not accepting yet", "red"))
            time.sleep(pause)


class APIClient:
    def __init__(self, port: int, host: str = "127.0.0.1", retry_seconds: float = 5.0, wire: str = "json"):
        self.host, self.port, self.retry_seconds = host, port, retry_seconds
        self.wire = wire                          # "pickle" to talk to an unmodified reference server (trusted host only)
        self.client_socket = None

    def run(self):
        self.client_socket = _dial(self.host, self.port, self.retry_seconds)
        greeting = self._recv()                   # the server greets every new connection
        _log.info(TextColors.colorize(f"connected: {greeting}", "cyan"))
        return greeting

    def _recv(self):
        return api_utils.receive_data(self.client_socket, allow_pickle=self.wire == "pickle")

    def get_output(self, **api_args):
        api_utils.send_data(self.client_socket, api_args, self.wire)
        return self._recv()

    def close(self):
        sock, self.client_socket = self.client_socket, None
        if sock is not None:
            api_utils.send_data(sock, {"terminate": True}, self.wire)
            sock.close()

    __enter__ = lambda self: (self.run(), self)[1]                  # noqa: E731

    def __exit__(self, *exc):
        self.close()
