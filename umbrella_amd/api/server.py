"""Single-engine TCP front end (surface of umbrella/api/server.py: APIServer(config, device, port, max_client).run()).

The engine serves one request at a time: reader threads (one per client) only enqueue requests, a single worker
thread owns the engine and answers them in arrival order."""
import queue
import socket
import threading

from ..logging_config import setup_logger
from ..utils import TextColors
from .api_utils import receive_data, send_data

_log = setup_logger()
_GREETING = {"status": "connected", "message": "Welcome to the server!"}


class APIServer:
    def __init__(self, config, device: str = "cuda:0", port: int = 65432, max_client: int = 4, host: str = "127.0.0.1",
                 engine=None, wire: str = "json"):
        self.config, self.device = config, device
        self.host, self.port, self.max_client = host, port, max_client
        self.engine = engine                      # a ready engine may be injected (tests)
        self.wire = wire                          # "pickle": serve unmodified reference clients (trusted host only)
        self.message_queue: queue.Queue = queue.Queue()
        self.queue_lock = threading.Lock()
        self._closing = threading.Event()
        self.server_socket = None

    # -- per-client reader -------------------------------------------------------------------------------
    def _requests(self, conn, addr):
        """Frames from one client until it says terminate, hangs up, or the server is closing."""
        while not self._closing.is_set():
            try:
                frame = receive_data(conn, allow_pickle=self.wire == "pickle")
            except Exception as err:              # closed socket / bad frame: drop this client only
                _log.error(TextColors.colorize(f"client {addr}: {err}", "red"))
                return
            if frame.get("terminate", False):
                return
            yield frame

    def handle_client(self, conn, addr):
        _log.info(TextColors.colorize(f"client {addr} connected", "cyan"))
        with conn:
            send_data(conn, _GREETING, self.wire)
            for request in self._requests(conn, addr):
                self.message_queue.put((addr, conn, request))
        _log.info(TextColors.colorize(f"client {addr} disconnected", "cyan"))

    # -- the only thread that touches the engine ------------------------------------------------------------
    def _validate(self, request: dict):
        """Wire input is untrusted: token ids must be ints inside the vocabulary before they reach the embedding gather."""
        ids = request.get("input_ids", None)
        if ids is None:
            return
        vocab = getattr(self.engine, "vocab_size", None)
        if not isinstance(ids, (list, tuple)) or not all(isinstance(i, int) and not isinstance(i, bool) for i in ids):
            raise ValueError("input_ids must be a list of integers")
        if vocab is not None and any(i < 0 or i >= vocab for i in ids):
            raise ValueError(f"input_ids outside the vocabulary [0, {vocab})")

    def _answer(self, request: dict) -> dict:
        with self.queue_lock:
            self._validate(request)
            result = self.engine.generate(**request)
        return dict(result, processed=True, response="Processed successfully")

    def process_queue(self):
        """One worker owns the engine.  A request that fails -- bad argument types, a kernel error, free text with the id
        tokenizer -- gets an error reply and the engine is reset; the worker (and every other client) carries on."""
        for addr, conn, request in iter(self.message_queue.get, None):
            try:
                reply = self._answer(request)
            except Exception as err:                                   # noqa: BLE001 -- per-request isolation
                _log.error(TextColors.colorize(f"request from {addr} failed: {type(err).__name__}: {err}", "red"))
                reply = {"processed": False, "response": f"{type(err).__name__}: {err}", "generated_text": "",
                         "generated_tokens": [], "avg_accept_tokens": 0, "time_per_output_token": 0}
                try:
                    with self.queue_lock:
                        self.engine.reset()
                except Exception as rerr:                               # noqa: BLE001
                    _log.error(TextColors.colorize(f"engine reset failed: {rerr}", "red"))
            try:
                send_data(conn, reply, self.wire)
            except (OSError, TypeError, ValueError) as err:
                _log.error(TextColors.colorize(f"reply to {addr} failed: {err}", "red"))

    def _ensure_engine(self):
        if self.engine is None:
            from ..speculation.auto_engine import AutoEngine
            self.engine = AutoEngine.from_config(self.device, **self.config)
            self.engine.initialize()

    def run(self):
        self._ensure_engine()
        listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        listener.bind((self.host, self.port))
        listener.listen(self.max_client)
        self.server_socket = listener
        threading.Thread(target=self.process_queue, daemon=True, name="umb-engine-worker").start()
        _log.info(TextColors.colorize(f"umbrella_amd server listening on {self.host}:{self.port}", "cyan"))
        while not self._closing.is_set():
            try:
                conn, addr = listener.accept()
            except OSError:                       # listener closed by shutdown()
                break
            threading.Thread(target=self.handle_client, args=(conn, addr), daemon=True).start()

    def shutdown(self):
        self._closing.set()
        if self.server_socket is not None:
            try:
                self.server_socket.close()
            except OSError:
                pass
