"""Wire framing kept from the reference (umbrella/api/api_utils.py:3-17):
4-byte big-endian length + pickle payload.  Pickle is only safe between mutually trusted
processes on one host -- the server binds 127.0.0.1 by default, as the reference does."""
import pickle


def send_data(conn, data):
    blob = pickle.dumps(data)
    conn.sendall(len(blob).to_bytes(4, "big"))
    conn.sendall(blob)


def receive_data(conn):
    head = b""
    while len(head) < 4:
        part = conn.recv(4 - len(head))
        if not part:
            raise ConnectionError("Connection lost while receiving data")
        head += part
    size = int.from_bytes(head, "big")
    buf = bytearray()
    while len(buf) < size:
        chunk = conn.recv(min(1024, size - len(buf)))
        if not chunk:
            raise ConnectionError("Connection lost while receiving data")
        buf += chunk
    return pickle.loads(bytes(buf))
