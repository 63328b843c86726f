"""Torch-free ensemble communicator: RCCL through the C ABI (include/quflow_hip.h, qf_comm_*).

SURVEY.md section 8e: replicas are independent, the only exchange is an all-gather of a few
float64 diagnostics per output chunk.  The reference has no distributed code, so there is no
reference interface to mirror; the object below offers the handful of torch.distributed calls
quflow_amd.ensemble and bench.py use (is_initialized / get_rank / get_world_size / barrier and
an all-gather of float64 rows), so either can be handed to them.

Bootstrap: ncclCommInitRank needs the same 128-byte id on every rank.  Rank 0 draws it and hands
it out over a plain TCP socket on (MASTER_ADDR, MASTER_PORT + 1) -- the launcher's own store
sits on MASTER_PORT -- to the world-1 ranks that connect; nothing is left behind on disk.
"""
import ctypes
import os
import socket
import time

import numpy as np

from . import _lib

ID_BYTES = 128


def exchange_id(rank, world, addr, port, make_id, timeout=120.0, nonce=None):
    """Rank 0 calls make_id() -> bytes and serves it to world-1 peers; the others fetch it.
    Pure host code (tests run it with a fake id and no GPU).  `nonce` (QUFLOW_COMM_NONCE, set by the
    launcher for all its ranks): a peer's hello is "<rank> <nonce>", and rank 0 ignores connections
    that do not carry it -- a stray or slow local connection can neither claim a rank id nor abort
    the hand-out (every per-connection failure is swallowed and the server keeps listening)."""
    if world == 1:
        return make_id()
    if nonce is None:
        nonce = os.environ.get("QUFLOW_COMM_NONCE", "")
    deadline = time.monotonic() + timeout
    if rank == 0:
        blob = make_id()
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            while True:           # the port may still be held by a previous owner for a moment
                try:
                    srv.bind((addr, port))
                    break
                except OSError:
                    if time.monotonic() > deadline:
                        raise
                    time.sleep(0.1)
            srv.listen(world)
            served = set()
            while len(served) < world - 1:
                srv.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise TimeoutError("id hand-out: %d of %d peers connected within %.0f s"
                                       % (len(served), world - 1, timeout))
                try:
                    with conn:
                        conn.settimeout(2.0)      # a peer says hello at once; a silent connection costs 2 s, not the hand-out
                        hello = conn.recv(96).decode("ascii", "replace").split()
                        if not hello or (nonce and (len(hello) < 2 or hello[1] != nonce)):
                            continue        # not one of ours
                        peer = int(hello[0])
                        if not 0 < peer < world:
                            continue
                        conn.sendall(blob)
                        served.add(peer)
                except (OSError, ValueError):
                    continue                # a stray, slow or broken connection: keep serving
        return blob
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                conn.sendall(("%d %s\n" % (rank, nonce)).encode("ascii"))
                blob = b""
                while len(blob) < ID_BYTES:
                    part = conn.recv(ID_BYTES - len(blob))
                    if not part:
                        break
                    blob += part
            if len(blob) == ID_BYTES:
                return blob
        except OSError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError("id hand-out: rank 0 at %s:%d not reachable within %.0f s" % (addr, port, timeout))
        time.sleep(0.05)


class NativeComm:
    """One RCCL communicator per process (= per GPU)."""

    def __init__(self, rank=None, world=None, device=None, addr=None, port=None, timeout=120.0):
        env = os.environ
        self.rank = int(env.get("RANK", 0)) if rank is None else int(rank)
        self.world = int(env.get("WORLD_SIZE", 1)) if world is None else int(world)
        self.device = int(env.get("LOCAL_RANK", 0)) if device is None else int(device)
        addr = addr or env.get("MASTER_ADDR", "127.0.0.1")
        # the hand-out's own port: QUFLOW_COMM_PORT when the launcher reserved one (bench.py does), else MASTER_PORT + 1
        if port is None:
            port = int(env["QUFLOW_COMM_PORT"]) if env.get("QUFLOW_COMM_PORT") else int(env.get("MASTER_PORT", 29500)) + 1
        port = int(port)
        self._lib = _lib.load()

        def make_id():
            buf = ctypes.create_string_buffer(ID_BYTES)
            _lib.check(self._lib.qf_comm_unique_id(buf))
            return buf.raw

        blob = exchange_id(self.rank, self.world, addr, port, make_id, timeout=timeout)
        handle = ctypes.c_void_p()
        _lib.check(self._lib.qf_comm_create(ctypes.byref(handle), self.device, self.world, self.rank,
                                             ctypes.c_char_p(blob)))
        self.handle = handle

    # the torch.distributed calls the ensemble code uses
    def is_initialized(self):
        return self.handle is not None

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def get_backend(self):
        return "rccl (native, qf_comm)"

    def barrier(self):
        _lib.check(self._lib.qf_comm_barrier(self.handle))

    def allgather_f64(self, values):
        """values: 1-d float64 of the same length on every rank -> (world, len) array."""
        send = np.ascontiguousarray(values, dtype=np.float64).ravel()
        recv = np.zeros((self.world, send.size), dtype=np.float64)
        if send.size:
            _lib.check(self._lib.qf_comm_allgather_f64(self.handle, send.ctypes.data_as(ctypes.c_void_p), send.size,
                                                       recv.ctypes.data_as(ctypes.c_void_p)))
        return recv

    def destroy_process_group(self):
        if self.handle is not None:
            self._lib.qf_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy_process_group()
        except Exception:
            pass
