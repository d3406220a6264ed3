#!/usr/bin/env python3
"""Client for bert-server (wire protocol of the reference: examples/sample_client.py:9-22).

    client.py [--port P] [--host H] [--texts FILE]      interactive nearest-text lookup
    client.py --port P --encode "some text"             print one embedding as JSON

On connect the server sends int32 n_embd; every sendall() of a text is answered by n_embd f32.
"""
import argparse
import json
import socket
import struct
import sys

import numpy as np


class EmbeddingClient:
    def __init__(self, host="127.0.0.1", port=8080):
        self.sock = socket.create_connection((host, port))
        self.n_embd = struct.unpack("<i", self._recv(4))[0]

    def _recv(self, n):
        chunks = []
        while n:
            c = self.sock.recv(n)
            if not c:
                raise ConnectionError("server closed the connection")
            chunks.append(c)
            n -= len(c)
        return b"".join(chunks)

    def encode(self, text: str) -> np.ndarray:
        data = text.encode()
        if not data:
            raise ValueError("an empty request closes the connection in this protocol")
        self.sock.sendall(data[: 1 << 15])
        return np.frombuffer(self._recv(4 * self.n_embd), dtype="<f4").copy()

    def close(self):
        self.sock.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8080)
    ap.add_argument("--texts", help="file with one text per line to search in")
    ap.add_argument("--encode", help="encode this text, print JSON, exit")
    ap.add_argument("-k", type=int, default=3)
    a = ap.parse_args()

    cl = EmbeddingClient(a.host, a.port)
    if a.encode is not None:
        print(json.dumps([float(x) for x in cl.encode(a.encode)]))
        return
    if not a.texts:
        sys.exit("--texts FILE or --encode TEXT required")
    texts = [t.strip() for t in open(a.texts, encoding="utf-8") if t.strip()]
    table = np.stack([cl.encode(t) for t in texts])
    print(f"Loaded {len(texts)} lines.")
    while True:
        try:
            q = input("Enter query: ")
        except EOFError:
            break
        if not q:
            break
        sims = table @ cl.encode(q)          # embeddings are L2-normalised: dot = cosine
        for i in np.argsort(-sims)[: a.k]:
            print(f"{sims[i]:.4f}  {texts[i]}")
    cl.close()


if __name__ == "__main__":
    main()
