"""The callers of the hot path, mapped onto streams and ranks: drop-ins for the reference's
`SlamSystem`, `AgentSystem` and `CloudSystem` (system/core.py:36-546) as far as they drive the path -- extraction, odometry,
mapping, loop closure, optimisation, the upload to the cloud and the cloud's multi-agent loop closure.  What the reference
hangs on the side of these classes (ResultLogger files and plots, ROS publishers, tqdm bars) is not here.

  SlamSystem.step(sensor_data)          core.py:360-423   one scan through extractor -> back end, exit code returned;
  SlamSystem.MT_Init/MT_Step/MT_Done/MT_Wait  core.py:102-358   the multi-thread mode.  The reference runs six threads over five
      queues and lets odometer, mapping and back end read and write ONE pose graph concurrently (the odometer of scan i + 1
      picks its partner from `last_known_keyframe` while the mapping thread of scan i is still deciding whether i is a
      key-frame: core.py:205-216 against mapping.py:180-195) -- its result depends on thread timing.  Here the mode keeps what
      is a pipeline and drops what is a race: an EXTRACTOR thread batches whatever has queued up (up to 32 scans, one launch
      chain, the sampling stage of the next batch on a side stream: extractor.MTExtractor.run) and a BACK-END thread takes the
      scans in order through odometry, mapping and loop closure on its own stream.  The trajectory equals `step`'s, scan for
      scan; the encoder overlaps the back end, which is where the time is.
  AgentSystem.start(dataloader) / wait  core.py:426-448   a thread feeding `step`; with a comm module every accepted key-frame
      is uploaded to member 0 (`UPLOAD_SCAN`: the scan, its odometry edge, its other edges);
  CloudSystem.start / wait / step       core.py:451-546   member 0: takes the uploads in arrival order, adds them to its graph
      and closes loops between agents.

`comm_module` is the reference's `Communicate_Module` interface: comm.RankCommunicateModule when agents and cloud are ranks
(tensors travel GPU to GPU), or any object with the same four methods (the reference's thread-shared dict of queues works
when everything lives in one process).
"""
from __future__ import annotations

import os
import threading
import time
from enum import Enum, unique
from queue import Queue
from typing import Dict, List, Optional

import numpy as np
import torch

from .consumer import ACPT, Rank0Consumer
from .extractor import MTExtractor
from .posegraph_optim import write_g2o


@unique
class EXIT_CODE(Enum):      # system/modules/utils.py:21-27
    acpt = 0
    drop = 10
    dist = 11
    engy = 12
    exit = 21


class ResultLogger:
    """The part of the reference's ResultLogger (system/modules/recoder.py:24-110) that pipeline/infer.py reads and that is
    text: stage timings (`record_perf` / `log_time` / `get_time_list`), the trajectory files (`save_trajectory`,
    recoder.py:76-97: every scan's and every key-frame's SE3_pred as twelve numbers per line, KITTI style, plus the step
    numbers) and the pose graph as g2o (`save_posegraph` -> PoseGraph.to_g2o_file, pose_graph.py:821-842).  The plots and the
    point-cloud map (`draw_trajectory`, `save_map`) are not provided: the calls are accepted and do nothing."""

    def __init__(self, backend: Rank0Consumer, log_dir: Optional[str]):
        self.backend, self.log_dir = backend, log_dir
        self.time_recorder: Dict[str, List[float]] = {}

    def record_perf(self, name: str, time_s: float) -> None:
        self.time_recorder.setdefault(name, []).append(time_s)

    def log_time(self, window: Optional[int] = None) -> dict:
        ret = {}
        for name, tl in self.time_recorder.items():
            t = [x for x in tl if x > 0.0] if window is None else (tl[-window:] if window < len(tl) else tl)
            ret[name] = (sum(t) / len(t), np.std(t))
        return ret

    def get_time_list(self, log_name: str) -> List[float]:
        return self.time_recorder[log_name].copy()

    def _path(self, name: str) -> str:
        if self.log_dir is None:
            raise ValueError("this SlamSystem was built without a logger_dir")
        os.makedirs(self.log_dir, exist_ok=True)
        return os.path.join(self.log_dir, name)

    def save_trajectory(self, file_name: str = "traj_kitti") -> None:
        b = self.backend
        scans = sorted(b.type, key=lambda t: t & 0xFFFF)          # by timestep, graph order among equals (stable)
        for kind, toks in (("all", scans), ("key", [t for t in scans if b.type[t] == "full"])):
            with open(self._path(f"{file_name}.{kind}frames.txt"), "w+") as f:
                for t in toks:
                    f.write(" ".join(f"{i:.10f}" for i in b.poses[t][:3, :].flatten().tolist()) + "\n")
            with open(self._path(f"{file_name}.{kind}steps.txt"), "w+") as f:
                for t in toks:
                    f.write(f"{int(t & 0xFFFF)}\n")

    def save_posegraph(self, file_name: str = "posegraph") -> None:
        b = self.backend
        edges = [(a, c, e["SE3"].double().numpy(), np.asarray(e["information"] if e["information"] is not None else np.eye(6)))
                 for (a, c), e in b.edges.items()]
        write_g2o(self._path(file_name + ".pg.g2o"), {t: b.poses[t].numpy() for t in b.type}, edges)

    def draw_trajectory(self, *a, **k) -> None:
        pass

    def save_map(self, *a, **k) -> None:
        pass


class SlamSystem:
    EXTRACTOR_BATCHSIZE = MTExtractor.EXTRACTOR_BATCHSIZE

    def __init__(self, args, dpm_encoder, dpm_decoder, system_id: int = 0, logger_dir: Optional[str] = None,
                 comm_module=None, device=None, keep_log: bool = False):
        self.args = args
        self.system_id = system_id
        self.coor_sys = system_id
        self.device = torch.device(device if device is not None else args.device)
        slam = dict(args.slam_system) if getattr(args, "slam_system", None) is not None else {}
        self.coor_scale = float(slam.get("coor_scale", 60))
        self.dpm_encoder, self.dpm_decoder = dpm_encoder, dpm_decoder
        self.extraction_thread = MTExtractor(dpm_encoder, coor_scale=self.coor_scale)
        # odometry_thread + mapping_thread + loop_thread + posegraph_map of the reference
        self.backend = Rank0Consumer(dpm_decoder, self.device, slam_args=slam, agent_id=system_id, keep_log=keep_log)
        self.result_logger = ResultLogger(self.backend, logger_dir if logger_dir is not None else getattr(args, "infer_tgt", None))
        self.comm_module = comm_module
        if comm_module is not None:
            self.comm_id = system_id
            comm_module.add_member(self.comm_id)
        self.codes: List[EXIT_CODE] = []
        self._mt = None

    # -- one scan ------------------------------------------------------------------------------------------------------
    def _full_pcd(self, point_cloud: torch.Tensor) -> torch.Tensor:
        """ScanPack.full_pcd (core.py:376): the scan in metres, (3,N) on the device"""
        return (point_cloud[:3].to(self.device, dtype=torch.float32) * self.coor_scale).contiguous()

    def _backend_step(self, desc: torch.Tensor, point_cloud: torch.Tensor) -> EXIT_CODE:
        t0 = time.perf_counter()
        tok, code = self.backend.step(desc, self._full_pcd(point_cloud))
        self.result_logger.record_perf("backend", time.perf_counter() - t0)      # odometer + mapping + loop_closure
        if code == ACPT and self.comm_module is not None:     # drop / dist leave step() before the upload (core.py:399-400)
            self.comm_module.send_message(caller=self.comm_id, callee=0, command="UPLOAD_SCAN",
                                          message=self.backend.upload_message(tok))
        out = EXIT_CODE[code]
        self.codes.append(out)
        return out

    @torch.no_grad()
    def step(self, sensor_data) -> EXIT_CODE:
        """sensor_data = [point_cloud (1,C,N) normalised, R, T, padding_mask (1,N), original_scan] (core.py:365)"""
        point_cloud, padding_mask = sensor_data[0], sensor_data[3]
        with torch.cuda.device(self.device):
            t0 = time.perf_counter()
            desc = self.extraction_thread.process(point_cloud=point_cloud, padding_mask=padding_mask)
            self.result_logger.record_perf("extract", time.perf_counter() - t0)   # enqueue time: the kernels run on
            return self._backend_step(desc[0], point_cloud[0])

    def trajectory(self):
        """(tokens, SE3_pred (n,4,4)) of every scan of the graph, by token -- what recoder.py:76-97 writes out"""
        toks = sorted(self.backend.poses)
        return toks, torch.stack([self.backend.poses[t] for t in toks])

    # -- multi-thread mode ---------------------------------------------------------------------------------------------
    def MT_Init(self):
        # the one device-side preparation of the mode: registrations replay captured graphs, and captures cannot happen once the
        # worker threads exist (Decoder.capture_registration_graphs) -- the odometer's pair and scan-to-map against 1 .. 16 scans
        if threading.active_count() == 1 and getattr(self.dpm_decoder, "graph_min_hits", 0) > 0:
            enc = self.args.encoder
            P, a = enc.npoint[len(enc.npoint) - 1 - enc.upsample_layers], self.backend.args   # descriptors per scan (256)
            self.dpm_decoder.capture_registration_graphs(
                [(P, P, a["registration_sample_odometer"])] + [(P * j, P, a["registration_sample_mapping"]) for j in range(1, 17)])
            # (one instance per shape: this mode's registrations all come from the back-end thread)
        q_in, q_mid = Queue(), Queue()
        errors: List[BaseException] = []
        end = object()      # behind the last scan (the reference forwards an exit code AHEAD of the scans it was drained with,
                            # core.py:147-151, and loses them when its downstream thread leaves on it)

        def guard(fn):
            def run():
                try:
                    with torch.cuda.device(self.device):
                        fn()
                except BaseException as e:  # noqa: BLE001 -- re-raised by MT_Wait
                    errors.append(e)
                    q_mid.put(end)
            return run

        def extractor():
            with torch.cuda.stream(torch.cuda.Stream(device=self.device)):
                self.extraction_thread.run(q_in, q_mid, make_scan=lambda item, d: (item, d),
                                           is_exit=lambda it: isinstance(it, EXIT_CODE), is_final=lambda it: it == EXIT_CODE.exit,
                                           to_host=False)
            q_mid.put(end)

        def backend():
            with torch.cuda.stream(torch.cuda.Stream(device=self.device)), torch.no_grad():
                while True:
                    it = q_mid.get()
                    if it is end:
                        return
                    if isinstance(it, EXIT_CODE):
                        continue
                    (item, d) = it
                    d.record_stream(torch.cuda.current_stream(self.device))
                    self._backend_step(d, item[1][0])

        t1, t2 = threading.Thread(target=guard(extractor), name="dpm-extractor"), threading.Thread(target=guard(backend), name="dpm-backend")
        self._mt = (q_in, (t1, t2), errors)
        t1.start(), t2.start()

    def MT_Step(self, sensor_data):
        point_cloud, R, T, padding_mask, original_scan = (list(sensor_data) + [None] * 5)[:5]
        self._mt[0].put((len(self.codes) / 10, point_cloud, R, T, padding_mask, original_scan))   # core.py:126-130

    def MT_Done(self):
        self._mt[0].put(EXIT_CODE.exit)

    def MT_Wait(self):
        _, threads, errors = self._mt
        for t in threads:
            t.join()
        self._mt = None
        if errors:
            raise RuntimeError("a thread of the multi-thread mode failed") from errors[0]


class AgentSystem(SlamSystem):
    def start(self, dataloader):
        def feed():
            try:
                for data in dataloader:
                    self.step(data)
            except BaseException as e:  # noqa: BLE001 -- re-raised by wait
                self._error = e
        self._error = None
        self._thread = threading.Thread(target=feed, name=f"dpm-agent-{self.system_id}")
        self._thread.start()

    def wait(self):
        self._thread.join()
        if self._error is not None:
            raise RuntimeError(f"agent {self.system_id} failed") from self._error


class CloudSystem(SlamSystem):
    def __init__(self, args, dpm_encoder, dpm_decoder, logger_dir: Optional[str] = None, comm_module=None, device=None,
                 keep_log: bool = False):
        assert comm_module is not None
        super().__init__(args, dpm_encoder, dpm_decoder, system_id=0, logger_dir=logger_dir, comm_module=comm_module,
                         device=device, keep_log=keep_log)
        self.communicate_module = comm_module
        self.arrivals: List[int] = []

    @torch.no_grad()
    def step(self, scan_pack: dict, odom_edge: Optional[dict], neighbor_edges: List[dict]):
        self.arrivals.append(scan_pack["token"])
        with torch.cuda.device(self.device):
            return self.backend.cloud_step(scan_pack, odom_edge, neighbor_edges)

    def _serve(self):
        while True:
            command, data = self.communicate_module.fetch_message(self.system_id, block=True)
            if command == "QUIT":
                return
            if command == "NO_OP" or (command == "AGENT_QUIT" and data is None):
                continue
            if command == "AGENT_QUIT":
                self.quit_agents.add(data)
                if self.expected_agents and len(self.quit_agents) >= self.expected_agents:
                    return
                continue
            if command != "UPLOAD_SCAN":
                raise RuntimeError(f"unknown operation code {command} to cloud {self.system_id}")
            self.step(scan_pack=data["new_scan"], odom_edge=data["odometer_edge"], neighbor_edges=data["neighbor_edges"])

    def start(self, expected_agents: int = 0):
        """expected_agents > 0: the loop also ends once that many agents sent AGENT_QUIT with their id (the reference's loop
        waits for a QUIT that its launcher sends, infer_multiagents.py)"""
        def serve():
            try:
                self._serve()
            except BaseException as e:  # noqa: BLE001 -- re-raised by wait
                self._error = e
        self._error, self.expected_agents, self.quit_agents = None, expected_agents, set()
        self._thread = threading.Thread(target=serve, name="dpm-cloud")
        self._thread.start()

    def wait(self):
        self._thread.join()
        if self._error is not None:
            raise RuntimeError("the cloud failed") from self._error
