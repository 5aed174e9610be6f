"""Length-prefixed message framing for the TCP API.

Frame = 4-byte big-endian payload length + payload (the reference's framing, umbrella/api/api_utils.py:3-17).
Payload encodings:

* ``json`` (default here): UTF-8 JSON object.  Requests / responses of the engines are plain dicts of str / int /
  float / list, so nothing else is needed, and a hostile peer cannot make the receiver execute code.
* ``pickle``: what the reference puts on the wire.  Only for talking to an unmodified reference client / server on a
  trusted host -- unpickling is code execution.  Receivers accept it only when ``allow_pickle=True``.

``receive_data`` recognises the encoding from the first payload byte (``{`` = JSON, 0x80 = pickle protocol >= 2).
"""
import json
import pickle

_MAX_FRAME = 256 << 20        # refuse absurd lengths instead of allocating them


def _encode(data, wire: str) -> bytes:
    if wire == "json":
        return json.dumps(data, separators=(",", ":")).encode("utf-8")
    if wire == "pickle":
        return pickle.dumps(data)
    raise ValueError(f"unknown wire encoding '{wire}'")


def send_data(conn, data, wire: str = "json"):
    blob = _encode(data, wire)
    conn.sendall(len(blob).to_bytes(4, "big") + blob)


def _read_exact(conn, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = conn.recv(min(1 << 16, n - len(buf)))
        if not chunk:
            raise ConnectionError("Connection lost while receiving data")
        buf += chunk
    return bytes(buf)


def receive_data(conn, allow_pickle: bool = False):
    size = int.from_bytes(_read_exact(conn, 4), "big")
    if size > _MAX_FRAME:
        raise ValueError(f"frame of {size} bytes exceeds the {_MAX_FRAME}-byte limit")
    blob = _read_exact(conn, size)
    if blob[:1] == b"{":
        return json.loads(blob.decode("utf-8"))
    if blob[:1] == b"\x80":
        if not allow_pickle:
            raise ValueError("pickle frame refused (pass allow_pickle=True only for a trusted reference peer)")
        return pickle.loads(blob)
    raise ValueError("unrecognised frame encoding")
