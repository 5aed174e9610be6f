"""Single-engine TCP server (umbrella/api/server.py:11-74): accept loop, one reader thread per
client, one worker draining a queue -- the engine is not thread-safe and serves one request at a time."""
import socket
import threading
from queue import Queue

from ..logging_config import setup_logger
from ..speculation.auto_engine import AutoEngine
from ..utils import TextColors
from .api_utils import receive_data, send_data

logger = setup_logger()


class APIServer:
    def __init__(self, config, device: str = "cuda:0", port: int = 65432, max_client: int = 4, host: str = "127.0.0.1",
                 engine=None):
        self.port, self.max_client, self.host, self.device, self.config = port, max_client, host, device, config
        self.engine = engine                      # tests may inject a ready engine
        self._stop = threading.Event()

    def handle_client(self, conn, addr):
        logger.info(TextColors.colorize(f"Connection from {addr}", "cyan"))
        try:
            send_data(conn, {"status": "connected", "message": "Welcome to the server!"})
            while True:
                try:
                    msg = receive_data(conn)
                    if msg.get("terminate", False):
                        break
                    self.message_queue.put((addr, conn, msg))
                except Exception as e:
                    logger.error(TextColors.colorize(f"Error handling data from {addr}: {e}", "red"))
                    break
        finally:
            conn.close()
            logger.info(TextColors.colorize(f"Connection with {addr} closed", "cyan"))

    def process_queue(self):
        while True:
            addr, conn, message = self.message_queue.get()
            with self.queue_lock:
                output = self.engine.generate(**message)
                reply = {**output, "processed": True, "response": "Processed successfully"}
                try:
                    send_data(conn, reply)
                except Exception as e:
                    logger.error(TextColors.colorize(f"Error sending data to {addr}: {e}", "red"))

    def run(self):
        if self.engine is None:
            self.engine = AutoEngine.from_config(self.device, **self.config)
            self.engine.initialize()
        self.server_socket = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.server_socket.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.server_socket.bind((self.host, self.port))
        self.server_socket.listen(self.max_client)
        logger.info(TextColors.colorize("umbrella_amd LLM server started successfully", "cyan"))
        self.message_queue = Queue()
        self.queue_lock = threading.Lock()
        threading.Thread(target=self.process_queue, daemon=True).start()
        while not self._stop.is_set():
            try:
                conn, addr = self.server_socket.accept()
            except OSError:
                break
            threading.Thread(target=self.handle_client, args=(conn, addr), daemon=True).start()

    def shutdown(self):
        self._stop.set()
        try:
            self.server_socket.close()
        except Exception:
            pass
