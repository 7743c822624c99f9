"""`Communicate_Module` of the reference's multi-agent mode (system/modules/utils.py:116-154) mapped onto ranks
("Communicate_Module -> per-GPU streams + RCCL p2p").

In the reference the agents and the cloud are THREADS of one process and the module is a dict of `queue.Queue`s:
`send_message(caller, callee, command, message)` puts `(command, message)` into the callee's queue,
`fetch_message(system_id, block)` takes the next one (non-blocking: `('NO_OP', None)` when empty), commands are
`NO_OP / UPLOAD_SCAN / AGENT_QUIT / QUIT`, and the only message with a payload is an agent's
`UPLOAD_SCAN` to the cloud (member 0): `dict(new_scan, odometer_edge, neighbor_edges)` -- descriptors (131,256), the scan
(3,N) and a few 4x4 / 6x6 matrices (core.py:411-422, consumed at core.py:529).

Here a member is a RANK (one process per GPU; member id == rank of the group) and the queue of a member lives in its
own process.  Same four methods, same return values.  A message travels in two parts:

  * control channel -- a gloo group: 16-byte preamble + the pickled message with every tensor replaced by a placeholder
    (shape, dtype).  Host-side, so a rank that is not expecting anything never parks a kernel on its GPU; one receiver
    thread per peer blocks in `recv` on it and fills the local queue, which is what makes `fetch_message(block=False)` and
    `get_queue_length` work exactly as in the reference;
  * data channel -- the tensors themselves, in placeholder order, point-to-point on `data_group`: RCCL over xGMI when that
    group's backend is nccl (device tensors go as they are, received on the receiver's current device), the control
    group itself otherwise (tensors staged through the host; a tensor that left a GPU arrives on the receiver's GPU
    when it has one).  The receives are posted only after the preamble announced them and the sends right after it, so
    both sides of an RCCL transfer always exist.  On the RCCL channel every call of this process -- the caller's sends, the
    receiver threads' receives -- takes ONE lock (a communicator must not be entered from two threads at once), and tensor
    payloads may only travel toward a LOWER member id: that is the reference's traffic (agents upload to the cloud, member
    0, core.py:411-422; the cloud answers with bare commands), and it makes the wait-for relation between blocked sends and
    receives acyclic -- two ranks that sent payloads to each other at the same moment would each hold their lock in a send
    whose receive the other cannot post.  A payload in the other direction raises; it has to go over a gloo data channel.

Per (sender, receiver) pair messages arrive in the order they were sent; between different senders the order is arrival
order, as with the reference's thread-fed queues.  `close()` ends the receiver threads (every member calls it: it is a
handshake on the control channel).
"""
from __future__ import annotations

import io
import pickle
import threading
from queue import Empty, Queue
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

OPERATIONS = ["NO_OP", "UPLOAD_SCAN", "AGENT_QUIT", "QUIT"]   # utils.py:118
_CLOSE = "__CLOSE__"                                           # receiver-thread shutdown (never handed to the caller)


class _TensorSlot:
    """placeholder of a tensor inside the pickled message"""
    __slots__ = ("index", "shape", "dtype", "was_cuda")

    def __init__(self, index, shape, dtype, was_cuda):
        self.index, self.shape, self.dtype, self.was_cuda = index, shape, dtype, was_cuda


class _Pickler(pickle.Pickler):
    def __init__(self, f, tensors):
        super().__init__(f, protocol=pickle.HIGHEST_PROTOCOL)
        self.tensors = tensors

    def persistent_id(self, obj):
        if isinstance(obj, torch.Tensor):
            self.tensors.append(obj)
            return ("dpm_tensor", len(self.tensors) - 1, tuple(obj.shape), obj.dtype, obj.is_cuda)
        return None


class _Unpickler(pickle.Unpickler):
    def __init__(self, f, tensors):
        super().__init__(f)
        self.tensors = tensors

    def persistent_load(self, pid):
        if pid[0] != "dpm_tensor":
            raise pickle.UnpicklingError("unknown persistent id")
        return self.tensors[pid[1]]


def _pack(command: str, message: Any) -> Tuple[bytes, List[torch.Tensor]]:
    tensors: List[torch.Tensor] = []
    buf = io.BytesIO()
    _Pickler(buf, tensors).dump((command, message))
    return buf.getvalue(), tensors


def _slots(blob: bytes) -> List[_TensorSlot]:
    """the tensor placeholders of a packed message, in order, without building the message"""
    found: List[_TensorSlot] = []

    class Scan(pickle.Unpickler):
        def persistent_load(self, pid):
            found.append(_TensorSlot(pid[1], pid[2], pid[3], pid[4]))
            return None
    Scan(io.BytesIO(blob)).load()
    return sorted(found, key=lambda s: s.index)


class RankCommunicateModule:
    OPERATIONS = OPERATIONS

    def __init__(self, control_group=None, data_group=None, device: Optional[torch.device] = None):
        """control_group: a gloo group over the members (default: a new gloo group over the world).  data_group: the group
        tensors travel on (default: the control group); pass the default nccl group for RCCL transfers.  device: where
        received device tensors are put (default: the current CUDA device, if any)."""
        if not dist.is_initialized():
            raise RuntimeError("RankCommunicateModule needs an initialised torch.distributed process group")
        self.control = control_group if control_group is not None else dist.new_group(backend="gloo")
        if dist.get_backend(self.control) != "gloo":
            raise ValueError("the control channel must be a gloo group (host-side, no kernels parked on the GPU)")
        self.data = data_group if data_group is not None else self.control
        self.data_is_nccl = dist.get_backend(self.data) == "nccl"
        self.rank, self.world = dist.get_rank(self.control), dist.get_world_size(self.control)
        self.device = device if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self.agents = set()
        self.logger: List[tuple] = []
        self.queue: "Queue[Tuple[str, Any]]" = Queue()
        self._send_lock = threading.Lock()      # one message at a time per process on the control channel
        self._nccl_lock = threading.Lock()      # EVERY RCCL call of this process (sends and receives), one at a time
        self._threads: Dict[int, threading.Thread] = {}
        self._closed = False
        self._errors: List[BaseException] = []
        for peer in range(self.world):
            if peer != self.rank:
                th = threading.Thread(target=self._receive_from, args=(peer,), daemon=True, name=f"dpm-comm-recv-{peer}")
                self._threads[peer] = th
                th.start()

    # -- the reference's interface (utils.py:126-154) ---------------------------------------------------------------
    def add_member(self, system_id: int) -> None:
        """utils.py:126-129.  Members are ranks; every rank of the group may be added on every rank (the reference's
        single module object is shared by all threads), only `system_id == rank` owns a queue here."""
        if not 0 <= system_id < self.world:
            raise ValueError(f"member {system_id} is not a rank of the group (world {self.world})")
        self.agents |= {system_id}

    def remove_member(self, system_id) -> None:
        self.agents.remove(system_id)

    def get_members(self):
        return list(self.agents)

    def send_message(self, caller: int, callee: int, command: str, message: Any):
        assert command in self.OPERATIONS
        assert caller in self.agents and callee in self.agents
        if caller != self.rank:
            raise ValueError(f"rank {self.rank} cannot send on behalf of member {caller}")
        self.logger.append((caller, callee, command, message))
        if callee == self.rank:                 # a member talking to itself: the reference's plain queue
            self.queue.put((command, message))
            return
        self._send(callee, command, message)

    def fetch_message(self, system_id, block=True):
        if system_id != self.rank:
            raise ValueError(f"rank {self.rank} holds the queue of member {self.rank}, not of {system_id}")
        self._raise_receiver_errors()
        if block:
            while True:
                try:
                    return self.queue.get(timeout=0.5)
                except Empty:
                    self._raise_receiver_errors()
        try:
            return self.queue.get_nowait()
        except Empty:
            return ("NO_OP", None)

    def get_queue_length(self, system_id):
        if system_id != self.rank:
            raise ValueError(f"rank {self.rank} holds the queue of member {self.rank}, not of {system_id}")
        return self.queue.qsize()

    # -- transport -----------------------------------------------------------------------------------------------
    def _send(self, callee: int, command: str, message: Any) -> None:
        blob, tensors = _pack(command, message)
        if tensors and self.data_is_nccl and callee >= self.rank:
            raise ValueError(f"tensor payloads on the RCCL data channel travel toward lower member ids only (member {self.rank} -> "
                             f"{callee}): see the module header; use a gloo data group for this message")
        payload = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        with self._send_lock:
            dist.send(torch.tensor([payload.numel(), len(tensors)], dtype=torch.int64), dst=self._global(callee), group=self.control)
            dist.send(payload, dst=self._global(callee), group=self.control)
            for t in tensors:
                t = t.detach().contiguous()
                if self.data_is_nccl:
                    if not t.is_cuda:
                        t = t.to(self.device)
                    with self._nccl_lock:
                        dist.send(t, dst=self._global(callee, self.data), group=self.data)
                else:
                    dist.send(t.cpu(), dst=self._global(callee), group=self.data)

    def loopback(self, t: torch.Tensor) -> torch.Tensor:
        """One tensor out through the data channel and back into this member: the send of `_send` and the receive of
        `_receive_from` (owned receive buffer on `self.device`, the process's one RCCL lock, the stream synchronised before
        the tensor is handed on) posted as ONE group, which is the only form in which a rank may talk to itself over RCCL.
        It is how a single-GPU box executes the channel at all (tests/test_gpu_rccl.py); members never need it."""
        t = t.detach().contiguous()
        me = self._global(self.rank, self.data)
        if self.data_is_nccl:
            src = t if t.is_cuda else t.to(self.device)
            got = torch.empty(src.shape, dtype=src.dtype, device=self.device)
            with self._nccl_lock:
                for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, me, group=self.data),
                                                   dist.P2POp(dist.irecv, got, me, group=self.data)]):
                    req.wait()
                torch.cuda.current_stream(self.device).synchronize()
            return got if t.is_cuda else got.cpu()
        got = torch.empty(t.shape, dtype=t.dtype)
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, t.cpu(), me, group=self.data),
                                           dist.P2POp(dist.irecv, got, me, group=self.data)]):
            req.wait()
        return got.to(self.device) if t.is_cuda and self.device.type == "cuda" else got

    def _receive_from(self, peer: int) -> None:
        try:
            while True:
                head = torch.empty(2, dtype=torch.int64)
                dist.recv(head, src=self._global(peer), group=self.control)
                payload = torch.empty(int(head[0]), dtype=torch.uint8)
                dist.recv(payload, src=self._global(peer), group=self.control)
                blob = payload.numpy().tobytes()
                tensors = []
                for slot in _slots(blob):
                    on_gpu = self.data_is_nccl or (slot.was_cuda and self.device.type == "cuda")
                    if self.data_is_nccl:
                        t = torch.empty(slot.shape, dtype=slot.dtype, device=self.device)
                        with self._nccl_lock:
                            dist.recv(t, src=self._global(peer, self.data), group=self.data)
                            torch.cuda.current_stream(self.device).synchronize()
                        if not slot.was_cuda:
                            t = t.cpu()
                    else:
                        t = torch.empty(slot.shape, dtype=slot.dtype)
                        dist.recv(t, src=self._global(peer), group=self.data)
                        if on_gpu:
                            t = t.to(self.device)
                    tensors.append(t)
                command, message = _Unpickler(io.BytesIO(blob), tensors).load()
                if command == _CLOSE:
                    return
                self.queue.put((command, message))
        except BaseException as e:  # noqa: BLE001 -- surfaced to the caller's thread by fetch_message / close
            if not self._closed:
                self._errors.append(e)

    def _global(self, group_rank: int, group=None) -> int:
        return dist.get_global_rank(group if group is not None else self.control, group_rank)

    def _raise_receiver_errors(self) -> None:
        if self._errors:
            raise RuntimeError(f"receiver thread of rank {self.rank} failed") from self._errors[0]

    def close(self) -> None:
        """Collective: every member tells every other one that nothing more will come, then joins its receivers."""
        if self._closed:
            return
        for peer in self._threads:
            blob, _ = _pack(_CLOSE, None)
            payload = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
            with self._send_lock:
                dist.send(torch.tensor([payload.numel(), 0], dtype=torch.int64), dst=self._global(peer), group=self.control)
                dist.send(payload, dst=self._global(peer), group=self.control)
        for th in self._threads.values():
            th.join(timeout=60)
        self._closed = True
        self._raise_receiver_errors()
