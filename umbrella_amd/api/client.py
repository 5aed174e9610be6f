"""Client side of the wire API (umbrella/api/client.py:7-35)."""
import socket
import time

from ..logging_config import setup_logger
from ..utils import TextColors
from .api_utils import receive_data, send_data

logger = setup_logger()


class APIClient:
    def __init__(self, port: int, host: str = "127.0.0.1", retry_seconds: float = 5.0):
        self.port, self.host, self.retry_seconds = port, host, retry_seconds

    def run(self):
        self.client_socket = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        while True:
            try:
                self.client_socket.connect((self.host, self.port))
                break
            except ConnectionRefusedError:
                logger.info(TextColors.colorize("Server is not available, retrying...", "red"))
                time.sleep(self.retry_seconds)
        hello = receive_data(self.client_socket)
        logger.info(TextColors.colorize(f"Server confirmation: {hello}", "cyan"))

    def get_output(self, **api_args):
        send_data(self.client_socket, api_args)
        return receive_data(self.client_socket)

    def close(self):
        send_data(self.client_socket, {"terminate": True})
        self.client_socket.close()
