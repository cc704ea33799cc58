"""Headline benchmark: images/sec of the SSD inference hot path (incl. NMS) on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus N ...          (N > 1, no launcher: starts N ranks itself, one per GPU, and relays rank 0's line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
(--gpus must equal the number of ranks the launcher started: a mismatch, or fewer devices than N, is an error, never a smaller run.)

A step = one pass of the whole hot path (stem .. heads .. softmax/decode/top-k/NMS/merge) over one device-resident
batch of synthetic 320x320 images (BASELINE.json configs[1]: ssdlite320_mobilenet_v3_large fp16, batch 64 per GPU).
Timing window = engine.evaluate's (engine.py:86-94): inputs already on the device, synchronize, forward incl.
post-process, outputs complete on the stream. The K timed steps go through demonet_amd.pipeline.ForwardPipeline: `--inflight`
(default 3) forwards are kept in flight, each a complete single-chain forward of its own batch on its own stream with its own
workspace and outputs -- the serving form of the path; `one_at_a_time` in the JSON line repeats the K steps with one forward in
flight (SSD.forward_batch) and `latency` is the per-forward distribution of that mode. N > 1: BASELINE configs[3] -- the global batch 256 is image-sharded, 256 / N per
rank (--batch overrides the per-GPU size: weak scaling); the fixed-shape
detections of every step are staged on the device and all-gathered over RCCL one window (16 steps) at a time, the last window
flushed inside the timed region (the reference gathers once, after the loop: engine.py:105).
Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16
DEFAULT_MODEL, DEFAULT_BATCH = "ssdlite320_mobilenet_v3_large", 64   # BASELINE.json configs[1]
C4_GLOBAL_BATCH = 256                                                  # BASELINE.json configs[3]: batch 256 over the GPUs of one node


def _family(name):
    """Kernel FAMILY = the template's name without its arguments: every template is grouped the same way (expdw_kernel<3,2,8,8,2,..> and
    expdw_kernel<3,1,8,16,..> are one family, like the per-reduction-length instantiations of pw_direct_kernel), launch-weighted.
    The dominant family is the one with the largest share of the step's kernel time."""
    name = name.replace(" ", "")
    cut = name.find("<")
    fam = name if cut < 0 else name[:cut]
    # expdw_one_kernel is the persistent form of expdw_kernel for the single-chunk blocks (round 4): the same stage of the network, one family
    return "expdw_kernel" if fam == "expdw_one_kernel" else fam


POINTWISE_FAMILIES = ("pw_direct_kernel", "pw_stream_kernel", "pw_wstat_kernel", "pw_kernel", "pw_xs_kernel", "pw_group_kernel")     # stand-alone 1x1 launches (north_star's ">= 90 % of roofline" path; the fused head launch, whose GEMM is a 1x1 behind a depthwise, is reported as its own family)


def softmax_levels(graph, kernel_names):
    """Pyramid levels whose softmax / decode / histogram run in the epilogue of the fused head launch (headfuse.hip, HeadFuseLevel::sm):
    told by what the library LAUNCHED -- dn_profile_op_info names that launch `head_fused_kernel<..,softmax>` -- not by re-reading a knob
    (round 5 read DN_HEAD_SOFTMAX with the wrong default and priced a full softmax launch that no longer exists). The epilogue takes the
    levels with at least 32 pixels per image; the rest leave as fp32 logits and get the small softmax_decode launch."""
    if not any(k.startswith("head_fused_kernel") and "softmax" in k for k in kernel_names):
        return ()
    return tuple(l for l, f in enumerate(graph.features) if graph.t(f).h * graph.t(f).w >= 32)


def op_costs(graph, n, sm_levels=()):
    """Algorithmic bytes / flops per op for a batch of n (fp16 tensors, fp32 bias; fused op = external bytes)."""
    out = []
    for nd in graph.nodes:
        ti, to = graph.t(nd.inp), graph.t(nd.out)
        if nd.op == "stem":
            m = n * to.h * to.w
            b = 4 * n * 3 * ti.h * ti.w + 2 * m * nd.cout + 4 * (27 * nd.cout + nd.cout)
            f = 2 * m * 3 * nd.k * nd.k * nd.cout
            kern = f"stem_kernel<{nd.cout}>"
        elif nd.op in ("pw", "conv"):
            m = n * to.h * to.w
            kk = nd.k * nd.k * nd.cin
            ob = 4 if nd.head else 2
            bi = 2 * m * (nd.cin if nd.op == "pw" else nd.cin * (ti.h * ti.w) / (to.h * to.w))
            bo = ob * m * nd.cout
            br = 2 * m * nd.cout if nd.residual >= 0 else 0         # the residual operand: read from HBM when it is not already an input of the launch
            bw = 2 * kk * nd.cout + 4 * nd.cout + br
            b = bi + bo + bw
            f = 2 * m * kk * nd.cout
            kern = "pw_kernel"
        elif nd.op == "dw":
            bi = 2 * n * ti.h * ti.w * nd.cin
            bo = 2 * n * to.h * to.w * nd.cin
            bw = 2 * nd.k * nd.k * nd.cin + 4 * nd.cin
            b = bi + bo + bw
            f = 2 * n * to.h * to.w * nd.cin * nd.k * nd.k
            kern = f"dw_kernel<{nd.k},{nd.stride}>"
        elif nd.op == "se":
            b = 4 * (2 * nd.cin * nd.squeeze + 2 * n * nd.cin)
            f = 4 * n * nd.cin * nd.squeeze
            kern = "se_fc_kernel"
        elif nd.op == "maxpool":
            b = 2 * n * (ti.h * ti.w + to.h * to.w) * nd.cin
            f = 0
            kern = "maxpool_kernel"
        else:
            b = 4 * n * ti.h * ti.w * nd.cin
            f = 3 * n * ti.h * ti.w * nd.cin
            kern = "l2norm_kernel"
        c = dict(kernel=kern, bytes=float(b), flops=float(f))
        if nd.op in ("pw", "dw"):
            c.update(b_in=float(bi), b_out=float(bo), b_w=float(bw), b_res=float(br) if nd.op == "pw" else 0.0)
        out.append(c)
    A, K = graph.num_anchors(), graph.num_classes
    topk, D = graph.post["topk_candidates"], graph.post["detections_per_img"]
    # softmax + decode launch: fp32 logits and regressions in, class-major scores and boxes out, one 1 KB histogram row per 64-anchor tile --
    # only for the anchors of the levels the fused head launch did NOT finish itself (sm_levels): with the epilogue on, 7 % of them
    a_done = sum(graph.t(graph.features[l]).h * graph.t(graph.features[l]).w * graph.anchors_per_loc[l] for l in sm_levels)
    a_left = A - a_done
    out.append(dict(kernel="softmax_decode_kernel", bytes=float(n * (a_left * (4 * K + 16 + 4 * (K - 1) + 16) + (a_left + 63) // 64 * 1024)), flops=float(5 * n * a_left * K)))
    out.append(dict(kernel="select_nms_kernel", bytes=float(n * (K - 1) * (4 * A + 24 * topk)), flops=float(25 * n * (K - 1) * topk * topk / 2)))
    out.append(dict(kernel="merge_kernel", bytes=float(n * ((K - 1) * topk * 8 + D * 40)), flops=0.0))
    out.append(dict(kernel="select_nms_kernel", bytes=float(4 * n), flops=0.0))      # the fallback launch behind the cut-off pass: reads the per-image flags (and redoes flagged images: usually none)
    return out


def fused_external_bytes(g, costs, mem, head, sm_levels=()):
    """External (HBM) algorithmic bytes of ONE fused launch covering ops `mem` ("a fused kernel is credited with the external
    bytes of the fused group")."""
    if head:
        # every level's two depthwise ops read the SAME feature map (once), their outputs never exist; the two 1x1 heads write fp32 rows:
        # logits [K] + regressions [4] per anchor -- or, for the levels whose post-process runs in the epilogue (sm_levels), scores [K - 1]
        # + decoded boxes [4] per anchor and one 1 KB histogram row per 32-pixel half tile and image (the logits never reach memory)
        K = g.num_classes
        ext = sum(costs[i]["b_in"] for i in mem if g.nodes[i].op == "dw") / 2 + sum(costs[i]["b_w"] for i in mem)
        for i in mem:
            nd = g.nodes[i]
            if nd.op != "pw":
                continue
            if nd.level in sm_levels and nd.head == 1:
                hw = g.t(nd.out).h * g.t(nd.out).w
                n = costs[i]["b_out"] / (4.0 * hw * nd.cout)
                ext += costs[i]["b_out"] * (K - 1) / K + n * ((hw + 30) // 32 + 1) * 1024
            else:
                ext += costs[i]["b_out"]
        return ext
    # the block's input once, its output once, every member's weights -- and a residual only when it is NOT the block's own input (an
    # inverted-residual block adds its input: those rows are already counted in first["b_in"]; round 4 counted them twice and
    # over-credited the fused-block family by 21 %)
    first, last = costs[mem[0]], costs[mem[-1]]
    inside = {g.nodes[mem[0]].inp} | {g.nodes[i].out for i in mem}          # a residual read from one of these never comes from HBM again
    ext = first["b_in"] + last["b_out"]
    for i in mem:
        c, nd = costs[i], g.nodes[i]
        if nd.op == "se":
            ext += 4 * 2 * nd.cin * nd.squeeze                                # the FC weights (the pooled vectors stay on chip)
        else:
            ext += c["b_w"] - (c["b_res"] if nd.residual in inside else 0.0)
    return ext


def cpu_baseline(name, graph, seed, budget_s=24.0):
    """The oracle (CPU restatement of the reference path, fp32 PyTorch eager) timed on this box's host cores: BASELINE
    config C1 (batch 1) and batch 8, each with torch.set_num_threads(1) -- what the reference's evaluation loop forces
    (engine.py:75) -- and with all usable cores; network (transform + backbone + heads + anchors) and post-process (softmax, decode,
    per-class top-k + NMS: python / numpy loops as in the reference) are timed separately. `value` = the fastest variant."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ssd_oracle as so
    import fast_post
    from demonet_amd import synth
    cores = os.cpu_count() or 1
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    many = min(cores, 16)          # more threads than this only adds oversubscription on these small convs
    o = so.OracleSSD(name, synth.state_dict(graph, 0), graph.num_classes, size=graph.size)
    W, H = graph.size
    hw = (H, W)
    imgs = [torch.from_numpy(synth.images(seed + i, 1, H, W)[0]) for i in range(8)]

    def run(batch, threads, budget, post="python"):
        torch.set_num_threads(threads)
        o(imgs[:1])                                   # warm-up
        pp = so.postprocess_detections if post == "python" else fast_post.postprocess_detections
        done, t_net, t_post = 0, 0.0, 0.0
        t0 = time.time()
        while True:
            a = time.time()
            r = o.forward_raw(imgs[:batch])
            b = time.time()
            dets = pp(r["cls_logits"], r["bbox_regression"], r["anchors"], hw, **o.post)
            for d, orig in zip(dets, r["orig"]):
                d["boxes"] = so.resize_boxes(d["boxes"], hw, orig)
            c = time.time()
            t_net += b - a
            t_post += c - b
            done += batch
            if time.time() - t0 > budget or done >= 64:
                break
        dt = time.time() - t0
        return {"batch": batch, "threads": threads, "postprocess": post, "images": done, "seconds": round(dt, 2), "images_per_sec": round(done / dt, 2),
                "network_ms_per_img": round(t_net / done * 1e3, 2), "postprocess_ms_per_img": round(t_post / done * 1e3, 2)}

    # two post-process implementations of the SAME function: "python" = the checker's numpy / Python loops (the port of the reference's
    # 90-iteration loop with a Python NMS: what this repo's oracle runs), "vectorised" = torch.topk per class + a compiled greedy NMS
    # (oracle/fast_post.py + nms_c.c: what the reference's loop costs on a box with torchvision's C++ nms). Six bounded variants.
    share = budget_s / 6.0
    variants = [run(8, many, share, "vectorised"), run(1, many, share, "vectorised"), run(1, 1, share, "vectorised"), run(8, 1, share, "vectorised"),
                run(8, many, share, "python"), run(1, 1, share, "python")]
    torch.set_num_threads(many)
    best = max(variants, key=lambda v: v["images_per_sec"])
    best_py = max((v for v in variants if v["postprocess"] == "python"), key=lambda v: v["images_per_sec"])
    net = min(variants, key=lambda v: v["network_ms_per_img"])
    return {"value": best["images_per_sec"], "unit": "images/sec", "cores": best["threads"], "host_cpus": cores, "kind": "port",      # cores = the threads the reported variant actually used
            "cpu_model": cpu_model,
            "postprocess": best["postprocess"],
            # the checker's own post-process (Python / numpy loops), for comparison with earlier rounds' lines:
            "port_python_postprocess_images_per_sec": best_py["images_per_sec"],
            # the network alone (transform + backbone + heads), fastest variant, for orientation:
            "network_only_images_per_sec": round(1e3 / net["network_ms_per_img"], 1),
            "sample": f"synthetic {H}x{W} images, full path incl. NMS, fp32 torch CPU eager, post-process = {best['postprocess']} "
                      f"(vectorised: torch.topk per class + compiled greedy NMS; python: the checker's loops); {cpu_model}, os.cpu_count() = {cores}; fastest of "
                      f"{len(variants)} bounded variants (batch {best['batch']}, {best['threads']} threads: {best['images']} images in "
                      f"{best['seconds']} s); per-variant split of network vs post-process time in `variants`",
            "variants": variants}


class _StubGraph:
    """--stub-cpu: the shapes bench.py needs from a model graph"""
    size = (320, 320)
    post = {"detections_per_img": 300, "topk_candidates": 300}
    nodes = ()


class _StubPipe:
    """--stub-cpu: stands in for demonet_amd.pipeline.ForwardPipeline on a box without a GPU (tests/test_dist_cpu.py runs main() with it at
    world size 2 over gloo, so the C4 path -- sharding, forwards in flight feeding the windowed gather, flush, barriers, MAX over ranks --
    is EXECUTED on CPU, not only its arithmetic). A 'forward' writes a payload that depends on the rank, the step and the shard."""

    def __init__(self, batch, dets, depth, rank, lo):
        self.B, self.D, self.depth, self.rank, self.lo, self.n = batch, dets, depth, rank, lo, 0
        self.bufs = [torch.zeros(batch, dets + 1, 6) for _ in range(depth)]
        self.joins = 0

    @staticmethod
    def payload(batch, dets, lo, step):
        p = torch.zeros(batch, dets + 1, 6)
        img = torch.arange(lo, lo + batch, dtype=torch.float32)
        p[:, :dets, 4] = (img[:, None] + 1) / 1024.0 + step              # scores encode (global image index, step)
        p[:, :dets, 5] = 1.0
        p[:, dets, 0] = (img % 7) + 1                                    # counts
        return p

    def submit(self, images, persistent_input=False):
        t = self.n
        self.bufs[t % self.depth].copy_(self.payload(self.B, self.D, self.lo, t))
        self.n += 1
        return t

    def packed(self, t):
        return self.bufs[t % self.depth]

    def stream_of(self, t):
        return None

    def join(self):
        self.joins += 1

    def result(self, t):
        p = self.bufs[t % self.depth]
        return p[:, :self.D, :4], p[:, :self.D, 4], p[:, :self.D, 5].long(), p[:, self.D, 0].int()

    def close(self):
        pass


def _launch_ranks(n, argv, stub):
    """`python bench.py --gpus N` (N > 1) with no launcher environment: start N fresh rank processes -- one per GPU, the reference's launch
    form (README.md:63 `python -m torch.distributed.launch --nproc_per_node=N`, util/misc.py:302-324 reads RANK / WORLD_SIZE / LOCAL_RANK) --
    as `python -m torch.distributed.run --nproc-per-node N bench.py ...`, relay rank 0's JSON line as the LAST line of stdout and return the
    children's status. The parent never initialises the GPU (torch.cuda.device_count() does not on this image) and never exec()s."""
    import socket
    import subprocess
    if not stub and os.environ.get("DN_BENCH_SHARE_GPU") != "1":
        have = torch.cuda.device_count()
        if have < n:
            sys.exit(f"bench.py: --gpus {n} but this node shows {have} GPU(s): refusing to print a {have}-GPU number as an {n}-GPU line")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if p.returncode != 0 or line is None:
        sys.exit(f"bench.py: the {n}-rank launch failed (status {p.returncode}, {'no' if line is None else 'a'} result line)")
    print(line, flush=True)
    return json.loads(line)


def time_config(model_name, batch, image_size, steps, warmup, dev, inflight=3):
    """One more single-GPU BASELINE config timed in this process with the loop of the headline (ForwardPipeline, `inflight` forwards in
    flight, device-resident synthetic batches, barrier-free N = 1 window: synchronize | K steps | synchronize). Used for BASELINE
    configs[2] (C3: ssd_lite_mobilenet_v2 at 300 x 300, batch 128) and configs[4] (C5: ssd512_vgg16, batch 32) behind the C2 sections."""
    from demonet_amd import models, synth
    from demonet_amd.pipeline import ForwardPipeline
    ncls = 21 if model_name == "ssd_lite_mobilenet_v2" else 91
    fkw = {"image_size": image_size} if (image_size and model_name == "ssd_lite_mobilenet_v2") else {}
    m = models.load_synthetic(getattr(models, model_name)(num_classes=ncls, **fkw), 0).to(dev)
    W, H = m.graph.size
    pipe = ForwardPipeline(m, batch, depth=inflight, chains=1, device=dev)
    batches = [torch.from_numpy(synth.images(5002 + j, batch, H, W)).to(dev) for j in range(inflight)]
    k = 0
    for _ in range(max(warmup, 2 * inflight)):
        pipe.submit(batches[k % inflight], persistent_input=True)
        k += 1
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        t = pipe.submit(batches[k % inflight], persistent_input=True)
        k += 1
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    counts = pipe.result(t)[3]
    out = {"value": round(batch * steps / dt, 1), "unit": "images/sec", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "warmup": max(warmup, 2 * inflight),
           "forwards_in_flight": inflight, "mean_detections": float(counts.float().mean().item()),
           "workload": f"{model_name} fp16, batch {batch}, {H}x{W} synthetic images, K={ncls}, synthetic weights seed 0, detections incl. NMS"}
    pipe.close()
    del pipe, m, batches
    torch.cuda.empty_cache()
    return out


def _on(stream):
    import contextlib
    return contextlib.nullcontext() if stream is None else torch.cuda.stream(stream)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--stub-cpu", action="store_true", help=argparse.SUPPRESS)      # CPU / gloo run of the multi-rank control path with a stub forward (tests)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: 64 on one GPU = BASELINE configs[1]; on N > 1 GPUs "
                                                     "the global batch 256 of configs[3] is sharded, 256 / N per GPU)")
    ap.add_argument("--model", default=DEFAULT_MODEL)
    ap.add_argument("--image-size", type=int, default=0, help="ssd_lite_mobilenet_v2 only: network input size (BASELINE config C3 is 300)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="plain launches instead of hipGraph replay")
    ap.add_argument("--input", choices=["f32", "u8"], default="f32",
                    help="f32: NCHW float images in [0,1] (the headline window, engine.py:86); u8: the decoder's [N,H,W,3] uint8 output")
    ap.add_argument("--weights", choices=["calibrated", "worstcase"], default="calibrated",
                    help="worstcase: class-head weights shrunk so every score passes score_thresh (all K-1 x topk candidates reach NMS; SURVEY 8d)")
    ap.add_argument("--inflight", type=int, default=-1, help="forwards kept in flight (demonet_amd.pipeline.ForwardPipeline: one stream, workspace and "
                    "output set per forward, each a single whole-batch chain). -1 = 3 (measured best of 2..12 at batch 32 and 64, tools/pipeline_probe.py), "
                    "1 = one forward at a time through SSD.forward_batch (two sub-batch chains per forward from 32 images up)")
    ap.add_argument("--chains", type=int, default=0, help="with --inflight 1 / --eager: sub-batch chains per forward (0 = the library's choice: 2 from 32 "
                    "images up). The PMC passes of tools/round_profile.sh use --eager --chains 1: the kernels at the launch size of the timed region")
    ap.add_argument("--no-latency", action="store_true", help="skip the per-step latency percentiles / D2H-inclusive step time (extra passes after the timed region)")
    ap.add_argument("--per-op", default="", help="write a per-op table (time, GB/s, TFLOP/s) to this file")
    ap.add_argument("--no-configs", action="store_true", help="skip the C3 / C5 legs (BASELINE configs[2] and [4]) that the default single-GPU run times behind the C2 sections")
    args = ap.parse_args(argv)
    stub = args.stub_cpu

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            # `python bench.py --gpus N` without a launcher: THIS process becomes the launcher (it has not touched the GPU and never will)
            return _launch_ranks(args.gpus, list(sys.argv[1:] if argv is None else argv), stub)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        # one rank per GPU (util/misc.py:302-324, README.md:63): a launcher that started a different number of ranks than --gpus names
        # would produce a line whose n_gpus contradicts the command
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch exactly one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ..., or plain `python bench.py --gpus {args.gpus}`)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    distributed = world > 1 or os.environ.get("DN_BENCH_FORCE_DIST") == "1"     # (the latter: 1-GPU smoke test of the RCCL path)
    share = False
    if stub:
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        # DN_BENCH_SHARE_GPU=1 (tests on a 1-GPU box): every rank uses cuda:0 and the collectives run over gloo -- the N-rank code path with
        # real forwards, not a throughput measurement (the line says so in config.parallelism)
        share = os.environ.get("DN_BENCH_SHARE_GPU") == "1"
        if share:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = lambda: torch.cuda.synchronize(dev)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")       # (only matters for the 1-GPU DN_BENCH_FORCE_DIST run; torchrun sets all of these)
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if stub:
            dist.init_process_group(backend="gloo")
        else:
            if share:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    from demonet_amd import models, synth
    from demonet_amd.dist import DetectionGatherer
    ncls = 21 if args.model == "ssd_lite_mobilenet_v2" else 91
    fkw = {"image_size": args.image_size} if (args.image_size and args.model == "ssd_lite_mobilenet_v2") else {}
    model = None if stub else models.load_synthetic(getattr(models, args.model)(num_classes=ncls, **fkw), 0)
    if args.weights == "worstcase":
        sd = {k: torch.from_numpy(v.copy()) for k, v in synth.state_dict(model.graph, 0).items()}
        for k in sd:
            last = (sd[k].dim() == 4 and sd[k].shape[1] > 1) or (sd[k].dim() == 1 and sd[k].shape[0] % ncls == 0 and k.endswith("bias"))
            if "classification_head" in k and last:
                sd[k] = sd[k] * 0.01          # near-uniform softmax: 1/K > score_thresh for every (anchor, class)
        model.load_state_dict(sd, strict=True)
    if not stub:
        model = model.to(dev)
    g = _StubGraph() if stub else model.graph
    W, H = g.size
    lo = 0
    if args.batch > 0:
        B, scaling = args.batch, "weak"
    elif world > 1:
        # BASELINE configs[3] (C4): global batch 256 image-sharded over the node, 256 / N contiguous images per rank
        # (dist.shard_range; the reference launches one process per GPU the same way: util/misc.py:302-324)
        from demonet_amd.dist import shard_range
        shard_lo = [shard_range(C4_GLOBAL_BATCH - C4_GLOBAL_BATCH % world, r, world)[0] for r in range(world)]
        lo, hi = shard_range(C4_GLOBAL_BATCH - C4_GLOBAL_BATCH % world, rank, world)      # equal shards (the gather is fixed-shape)
        B, scaling = hi - lo, "strong"
    else:
        B, scaling = DEFAULT_BATCH, "weak"
    if scaling == "weak":
        lo = rank * B
    images = torch.zeros(1) if stub else torch.from_numpy(synth.images(1002 + rank, B, H, W)).to(dev)      # device-resident input (engine.py:86)
    if args.eager and not stub:
        model.set_graph_mode(False)
    to_u8 = lambda x: (x * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    images_u8 = to_u8(images) if args.input == "u8" else None

    gatherer = DetectionGatherer(B, g.post["detections_per_img"], dev) if distributed else None
    R = args.inflight if args.inflight > 0 else 3
    if args.eager:
        R = 1
    pipe = None
    if R == 1 and args.chains > 0:
        from demonet_amd import _lib as _l
        _l.check(_l.lib().dn_set_chains(C.c_void_p(model._plan(dev)), args.chains))
    if stub:
        pipe = _StubPipe(B, g.post["detections_per_img"], R, rank, lo)
        batches = [images] * R
    elif R > 1:
        # the serving form: R forwards in flight, each a single whole-batch chain (demonet_amd/pipeline.py); one device-resident
        # synthetic batch per slot (slot 0 holds the batch of the one-at-a-time mode)
        from demonet_amd.pipeline import ForwardPipeline
        pipe = ForwardPipeline(model, B, depth=R, chains=1, device=dev, uint8=images_u8 is not None, packed=distributed)
        batches = [images] + [torch.from_numpy(synth.images(3002 + 16 * rank + j, B, H, W)).to(dev) for j in range(1, R)]
        if images_u8 is not None:
            batches = [images_u8] + [to_u8(x) for x in batches[1:]]
    nstep = [0]
    last_ticket = [-1]

    def step():
        if pipe is not None:
            k = nstep[0]
            nstep[0] += 1
            t = pipe.submit(batches[k % R], persistent_input=True)
            if distributed:
                # the merge kernel writes the gather payload itself; it is staged per step (on the forward's own stream) and
                # all-gathered per window of steps
                with _on(pipe.stream_of(t)):
                    last_ticket[0] = gatherer.submit(src=pipe.packed(t), join=pipe.join)
            return t
        if distributed:
            boxes, scores, labels, counts = model.forward_batch(images, persistent_input=True, packed=gatherer.next_buffer())
            gatherer.submit()
        elif images_u8 is not None:
            boxes, scores, labels, counts = model.forward_uint8(images_u8)
        else:
            boxes, scores, labels, counts = model.forward_batch(images, persistent_input=True)
        return counts

    for _ in range(max(args.warmup, 2 * R)):
        step()
    sync()
    if distributed:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counts = step()
    if distributed:
        if pipe is not None:
            with _on(pipe.stream_of(counts)):
                gatherer.flush(join=pipe.join)
        else:
            gatherer.flush()
    sync()
    if distributed:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt          # whole-job aggregate: every rank processed B images per step
    if stub:
        # what the last step's gather holds on every rank: the global batch in image order, every rank's shard where shard_range put it
        total_steps = max(args.warmup, 2 * R) + args.steps
        pk, cn = gatherer.result(last_ticket[0])
        D = g.post["detections_per_img"]
        per = pk.shape[0] // world
        ok = pk.shape[0] == world * B
        for r in range(world):
            want = _StubPipe.payload(B, D, r * B if scaling == "weak" else shard_lo[r], total_steps - 1)
            ok = ok and torch.equal(pk[r * per:(r + 1) * per], want[:, :D]) and torch.equal(cn[r * per:(r + 1) * per], want[:, D, 0].int())
        result = {"metric": "stub", "value": round(value, 1), "n_gpus": world, "rccl_ranks_seen": dist.get_world_size(), "steps": args.steps, "ms_per_step": round(ms_per_step, 4), "scaling": scaling,
                  "global_batch": B * world, "per_rank_batch": B, "shard": [lo, lo + B], "gather_windows": gatherer.gathered // gatherer.K,
                  "joins": pipe.joins, "stub_check": bool(ok)}
        dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(result), flush=True)
        return result
    if pipe is not None:
        counts = pipe.result(counts)[3]
        chains_timed = 1
        pipe.close()                              # back to the automatic split: the sections below run one forward at a time
        pipe = None
    else:
        chains_timed = model.batch_split(B)

    result = {
        "metric": "images/sec ssdlite320_mobilenet_v3_large fp16 end-to-end incl. NMS" if args.model.startswith("ssdlite320")
                  else f"images/sec {args.model} fp16 end-to-end incl. NMS",
        "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "ms_per_img": round(ms_per_step / B, 5),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "rccl_ranks_seen": dist.get_world_size() if distributed else 1,        # what the communicator itself reports after init (1: no communicator)
        "dtype": "fp16", "data": "synthetic",
        "config": {"workload": (f"{args.model} fp16, global batch {B * world} image-sharded over {world} GPUs ({B} per GPU), " if scaling == "strong"
                                else f"{args.model} fp16, batch {B} per GPU, ") + f"{H}x{W} synthetic images, K={ncls}, "
                               f"synthetic weights seed 0, post-process incl. per-class top-{g.post['topk_candidates']} + hard NMS",
                   "global_batch": B * world, "forwards_in_flight": R,
                   "launch": (f"hipGraph replay, {R} forwards in flight (one stream, workspace and output set each; one plan), each forward a single chain of {B} images" if R > 1 else
                              ("eager" if args.eager else "hipGraph replay") + (f", one forward at a time, {model.batch_split(B)} sub-batch chains" + (
                                  " as branches of one graph" if B >= 64 or os.environ.get("DN_CHAIN_GRAPHS") == "0" else " as one graph each on its own stream") if model.batch_split(B) > 1 else ", one forward at a time, one chain")),
                   "parallelism": (f"image-sharded x{world}, RCCL all_gather of detections (windows of {gatherer.K} steps)" +
                                   (" -- DN_BENCH_SHARE_GPU: every rank on cuda:0 over gloo, a code-path check, NOT a scaling number" if os.environ.get("DN_BENCH_SHARE_GPU") == "1" else "")) if distributed else "single GPU",
                   "input": "NCHW fp32 in [0,1], device-resident" if args.input == "f32" else "NHWC uint8 (decoder output), device-resident",
                   "mean_detections": float(counts.float().mean().item())},
    }
    if args.weights != "calibrated":
        result["config"]["workload"] += f", {args.weights} class-head weights"

    if rank == 0 and not args.no_latency and not distributed:
        # per-step distribution (one forward in flight at a time) and the window of engine.py:86-94 with the
        # detections copied to the host inside it. Extra passes after the timed region; `value` is unaffected.
        lat = []
        for _ in range(max(args.steps, 50)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step()
            e1.record()
            e1.synchronize()
            lat.append(e0.elapsed_time(e1))
        lat.sort()
        pick = lambda q: round(lat[min(len(lat) - 1, int(q * len(lat)))], 4)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            if images_u8 is not None:
                out = model.forward_uint8(images_u8)
            else:
                out = model.forward_batch(images, persistent_input=True)
            host = [t.cpu() for t in out]
        d2h_ms = (time.perf_counter() - t1) / args.steps * 1e3
        if R > 1:
            # the same K steps one forward at a time (SSD.forward_batch, its own in-forward split): what the pipelining buys
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize(dev)
            dt2 = time.perf_counter() - t2
            result["one_at_a_time"] = {"value": round(B * args.steps / dt2, 1), "unit": "images/sec", "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                                       "sub_batch_chains": model.batch_split(B)}
        # what a demonet user gets from `model(images)` (engine.py:86-94: a list of [3, H, W] images in, a list of dicts out, one batch at a
        # time): stacking the list, the forward, ONE host sync for the counts, the per-image views
        list_ms = None
        if images_u8 is None:
            img_list = list(images.unbind(0))
            model(img_list)
            torch.cuda.synchronize(dev)
            ll = []
            for _ in range(max(args.steps // 4, 20)):
                t3 = time.perf_counter()
                dets = model(img_list)
                torch.cuda.synchronize(dev)
                ll.append((time.perf_counter() - t3) * 1e3)
            ll.sort()
            list_ms = round(ll[len(ll) // 2], 4)
        result["latency"] = {"p10_ms": pick(0.10), "median_ms": pick(0.50), "p90_ms": pick(0.90), "samples": len(lat),
                             "list_api_median_ms": list_ms,
                             "ms_per_step_with_d2h": round(d2h_ms, 4),
                             "d2h_bytes": int(sum(t.numel() * t.element_size() for t in host))}

    if rank == 0 and not args.no_roofline:
        # per-kernel device time: HIP events around every launch on the forward stream (eager pass, same inputs)
        from demonet_amd import _lib
        L = _lib.lib()
        h = C.c_void_p(model._handle)
        nseg = len(g.nodes) + 4
        if chains_timed == 1 and _lib.check(L.dn_batch_split(h, B)) != 1:
            # the timed region ran whole-batch chains: profile the kernels at that size
            _lib.check(L.dn_set_chains(h, 1))
            model._bufs = {}
        _lib.check(L.dn_profile_begin(h))
        for _ in range(min(args.steps, 20)):
            model.forward_batch(images, persistent_input=True)
        buf = (C.c_float * nseg)()
        runs = _lib.check(L.dn_profile_end(h, buf, nseg))
        split = _lib.check(L.dn_batch_split(h, B))     # every kernel runs once per sub-batch branch
        # the library reports which kernel each op launched and which op's event segment holds a grouped launch's time
        name = C.create_string_buffer(96)
        owner = C.c_int32()
        # the post-process launches have event segments of their own (dn_profile_end): softmax + decode (with the fused head launch: only
        # the small levels, and the cut-off in its last tile per image) | [tau +] cut-off selection | merge | the fallback launch
        launched = []
        for i in range(len(g.nodes)):
            _lib.check(L.dn_profile_op_info(h, i, name, 96, C.byref(owner)))
            launched.append(name.value.decode())
        sm_levels = softmax_levels(g, launched)
        costs = op_costs(g, B, sm_levels)
        post = [costs[len(g.nodes)]["kernel"], "select_nms_fast_kernel (+tau when it is a launch)", "merge_kernel", "select_nms_kernel (fallback)"]
        agg = {}
        for i, c in enumerate(costs):
            if i < len(g.nodes):
                _lib.check(L.dn_profile_op_info(h, i, name, 96, C.byref(owner)))
                c["kernel"], c["owner"] = name.value.decode(), owner.value
            else:
                c["kernel"], c["owner"] = post[i - len(g.nodes)], i
        # fused inverted-residual launches: the intermediate activations never reach HBM -> external bytes only
        fused = {}
        for i, c in enumerate(costs):
            if c["kernel"].startswith(("expdw_kernel", "expdw_one_kernel", "pw_dw_direct_kernel", "head_fused_kernel")):
                fused.setdefault(c["owner"], []).append(i)
        for mem in fused.values():
            ext = fused_external_bytes(g, costs, mem, costs[mem[0]]["kernel"].startswith("head_fused_kernel"), sm_levels)
            for i in mem:
                costs[i]["bytes"] = ext / len(mem)
        owners = {}
        for i, c in enumerate(costs):
            if c["kernel"].startswith("("):
                continue                            # no launch of its own (a squeeze-excitation computed inside another launch): its event segment is empty
            a = agg.setdefault(_family(c["kernel"]), dict(ms=0.0, bytes=0.0, flops=0.0, launches=0, members=set()))
            a["bytes"] += c["bytes"]
            a["flops"] += c["flops"]
            a["members"].add(c["kernel"])
            if c["owner"] not in owners:
                owners[c["owner"]] = c["kernel"]
                a["ms"] += buf[c["owner"]]          # summed over the sub-batch launches of one forward
                a["launches"] += split
        if args.per_op:
            with open(args.per_op, "w") as f:
                names = [f"{nd.op}:{nd.conv_key or nd.fc1_key or nd.scale_key}" for nd in g.nodes] + ["softmax_decode (small levels + cut-off)", "select_nms_fast", "merge", "fallback select"]
                members = {}
                for i, c in enumerate(costs):
                    members.setdefault(c["owner"], []).append(i)
                for i, (c, nm) in enumerate(zip(costs, names)):
                    nd = g.nodes[i] if i < len(g.nodes) else None
                    shp = f"{g.t(nd.inp).c}x{g.t(nd.inp).h}x{g.t(nd.inp).w}->{g.t(nd.out).c}x{g.t(nd.out).h}x{g.t(nd.out).w} k{nd.k}s{nd.stride}" if nd else ""
                    if len(members[c["owner"]]) > 1:      # member of a grouped launch: timed as a whole below
                        f.write(f"{i:3d} {nm:58s} {shp:34s}  (slot {c['owner']:3d}) {c['bytes'] / 1e6:8.1f} MB {'':26s}  {c['kernel']}\n")
                        continue
                    ms = buf[c["owner"]]          # a launch's time sits in its segment slot, which need not be the op's own index
                    f.write(f"{i:3d} {nm:58s} {shp:34s} {ms * 1e3:8.1f} us {c['bytes'] / 1e6:8.1f} MB {c['bytes'] / max(ms, 1e-9) / 1e6:8.0f} GB/s {c['flops'] / max(ms, 1e-9) / 1e9:7.1f} TF/s  {c['kernel']}\n")
                for slot, mem in sorted(members.items()):
                    if len(mem) < 2:
                        continue
                    ms = buf[slot]
                    gb = sum(costs[i]["bytes"] for i in mem)
                    fl = sum(costs[i]["flops"] for i in mem)
                    f.write(f"grouped launch, slot {slot:3d}: ops {mem}  {ms * 1e3:8.1f} us {gb / 1e6:8.1f} MB {gb / max(ms, 1e-9) / 1e6:8.0f} GB/s {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s  {costs[mem[0]]['kernel']}\n")
        total_ms = sum(a["ms"] for a in agg.values())
        dom = max(agg, key=lambda k: agg[k]["ms"])        # by the live event pass; replaced below by the dominant family of the TIMED command's rocprofv3 summary when one is committed

        def roof(x, ms):
            # which roof bounds it: arithmetic intensity against the machine balance (2500 TF/s / 8 TB/s = 312 flop/B)
            mfma = x["flops"] / max(x["bytes"], 1.0) > MFMA_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS
            if mfma:
                return "mfma", x["flops"] / (ms * 1e-3) / 1e12, MFMA_PEAK_TFLOPS, "TFLOP/s"
            return "hbm", x["bytes"] / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"

        import csv, glob, re
        prof_tag = {("ssdlite320_mobilenet_v3_large", 64): "", ("ssdlite320_mobilenet_v3_large", 32): "_batch32", ("ssd512_vgg16", 32): "_vgg512",
                    ("ssd300_vgg16", 64): "_vgg300", ("ssd_lite_mobilenet_v2", 128): "_v2_300" if H == 300 else None}.get((args.model, B))

        def latest(suffix):
            if prof_tag is None:
                return None
            paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]{prof_tag}_{suffix}")), reverse=True)
            return paths[0] if paths else None

        def traffic_of(fam_name):
            # HBM bytes per launch from the PMC passes committed under profiles/ (tools/round_profile.sh + tools/pmc_traffic.py), launch-weighted
            path = latest("hbm_traffic.json")
            if not path:
                return None, None
            with open(path) as f:
                tk = json.load(f)["kernels"]
            tot, cnt = 0.0, 0
            for k, v in tk.items():
                if _family(k) == fam_name:
                    tot += v["hbm_bytes_per_launch"] * v.get("launches_sampled", 1)
                    cnt += v.get("launches_sampled", 1)
            return (round(tot / cnt), os.path.relpath(path, ROOT)) if cnt else (None, None)

        def rocprof_families(suffix):
            # per-family totals of a committed rocprofv3 --kernel-trace --stats summary (tools/round_profile.sh): "kernel_stats.csv" = this
            # command (graph replay, forwards in flight: a launch shares the chip), "kernel_stats_one_forward.csv" = --inflight 1 --chains 1
            # (one single-chain forward at a time: the condition of the live event pass, without its ~3-5 us eager dispatch gap per launch)
            path = latest(suffix)
            if not path:
                return None, None
            fam = {}
            with open(path) as f:
                for r in csv.DictReader(f):
                    nm = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
                    nm = re.sub(r"\(.*$", "", nm).replace("void ", "").strip()
                    if nm.startswith("__amd") or "at::" in nm:
                        continue
                    e = fam.setdefault(_family(nm), [0.0, 0])
                    e[0] += float(r["TotalDurationNs"])
                    e[1] += int(r["Calls"])
            return fam, os.path.relpath(path, ROOT)

        fam_t, src_t = rocprof_families("kernel_stats.csv")
        fam_1, src_1 = rocprof_families("kernel_stats_one_forward.csv")
        dom_live = dom
        if fam_t:
            # the dominant kernel family = the largest share of the kernel time of the timed command itself (graph replay, forwards in flight),
            # from the committed summary of this configuration; its duration below is still measured live (HIP events)
            ranked = [k for k in sorted(fam_t, key=lambda k: -fam_t[k][0]) if k in agg]
            if ranked:
                dom = ranked[0]
        d = agg[dom]
        bound, ach, peak, unit = roof(d, d["ms"])
        traffic, traffic_src = traffic_of(dom)
        rp = {}
        if fam_1 and dom in fam_1:
            avg_us = fam_1[dom][0] / fam_1[dom][1] / 1e3
            _, ach1, _, _ = roof(d, avg_us * 1e-3 * d["launches"])
            rp = {"source": src_1, "avg_launch_us": round(avg_us, 2), "achieved": round(ach1, 1), "frac": round(ach1 / peak, 4),
                  "share_of_kernel_time": round(fam_1[dom][0] / sum(v[0] for v in fam_1.values()), 3)}
        if fam_t:
            tot_t = sum(v[0] for v in fam_t.values())
            dom_t = max(fam_t, key=lambda k: fam_t[k][0])
            rp.update({"timed_command_source": src_t, "timed_command_dominant": dom_t, "timed_command_dominant_share": round(fam_t[dom_t][0] / tot_t, 3)})
            if dom in fam_t:
                rp.update({"timed_command_avg_launch_us": round(fam_t[dom][0] / fam_t[dom][1] / 1e3, 2), "timed_command_share": round(fam_t[dom][0] / tot_t, 3)})
        # the whole step against the HBM roof: external algorithmic bytes of every launch / the TIMED step
        step_bytes = sum(a["bytes"] for a in agg.values())
        step_flops = sum(a["flops"] for a in agg.values())
        # the stand-alone 1x1 launches together (north_star: ">= 90 % of fp16 roofline on the pointwise-conv hot path"); the 1x1 convs inside
        # the fused block kernels (expdw_kernel, pw_dw_direct_kernel) are accounted with those families
        pw = dict(ms=sum(agg[k]["ms"] for k in agg if k in POINTWISE_FAMILIES), bytes=sum(agg[k]["bytes"] for k in agg if k in POINTWISE_FAMILIES),
                  flops=sum(agg[k]["flops"] for k in agg if k in POINTWISE_FAMILIES), launches=sum(agg[k]["launches"] for k in agg if k in POINTWISE_FAMILIES))
        pw_obj = None
        if pw["ms"] > 0:
            # per-launch roofline time max(bytes / 8 TB/s, flops / 2.5 PF/s) summed over the launches, against the measured time
            t_roof_ms = 0.0
            seen = set()
            for i, c in enumerate(costs):
                if not c["kernel"].startswith("(") and _family(c["kernel"]) in POINTWISE_FAMILIES:
                    t_roof_ms += max(c["bytes"] / (HBM_PEAK_GBS * 1e9), c["flops"] / (MFMA_PEAK_TFLOPS * 1e12)) * 1e3
            pw_obj = {"families": [k for k in agg if k in POINTWISE_FAMILIES], "launches_per_step": pw["launches"], "ms": round(pw["ms"], 4),
                      "GB/s": round(pw["bytes"] / pw["ms"] / 1e6, 1), "TFLOP/s": round(pw["flops"] / pw["ms"] / 1e9, 1),
                      "roofline_ms": round(t_roof_ms, 4), "frac": round(t_roof_ms / pw["ms"], 4)}
            if fam_1:
                ns = sum(fam_1[k][0] / max(fam_1[k][1], 1) * agg[k]["launches"] for k in agg if k in POINTWISE_FAMILIES and k in fam_1)
                if ns > 0:
                    pw_obj["rocprof_ms"] = round(ns / 1e6, 4)
                    pw_obj["rocprof_frac"] = round(t_roof_ms / (ns / 1e6), 4)
        result["roofline"] = {"kernel": dom, "dominant_by": "share of the timed command's kernel time (committed rocprofv3 summary)" if fam_t else "share of the live event pass",
                              "dominant_of_live_event_pass": dom_live, "members": sorted(d["members"]), "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit,
                              "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                              "timing": "HIP events around every launch of one eager single-chain forward on the launch stream (includes the eager dispatch gap); `rocprof` = the same family in the committed rocprofv3 summaries",
                              "launches_per_step": d["launches"], "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2),
                              "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                              "algorithmic_flops_per_launch": round(d["flops"] / d["launches"]),
                              "share_of_step": round(d["ms"] / total_ms, 3), "profiled_runs": runs,
                              "eager_sum_ms": round(total_ms, 4), "rocprof": rp or None,
                              "step": {"algorithmic_bytes": round(step_bytes), "algorithmic_flops": round(step_flops), "ms_per_step": round(ms_per_step, 4),
                                       "GB/s": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "TFLOP/s": round(step_flops / (ms_per_step * 1e-3) / 1e12, 1)},
                              "pointwise": pw_obj}
        result["kernels"] = {k: {"ms": round(v["ms"], 4), "GB/s": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1),
                                 "TFLOP/s": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2), "launches": v["launches"],
                                 "share": round(v["ms"] / total_ms, 3)}
                             for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    if rank == 0 and world == 1 and not distributed and not args.no_configs and args.model == DEFAULT_MODEL and B == DEFAULT_BATCH and args.weights == "calibrated":
        # the other single-GPU BASELINE configs, same loop, same process (bounded: K steps each): C3 = configs[2], C5 = configs[4]
        model = None
        torch.cuda.empty_cache()
        result["configs"] = {"C2": {"value": result["value"], "unit": "images/sec", "ms_per_step": result["ms_per_step"], "steps": args.steps, "warmup": args.warmup,
                                    "forwards_in_flight": R, "workload": result["config"]["workload"]},
                             "C3": time_config("ssd_lite_mobilenet_v2", 128, 300, args.steps, args.warmup, dev),
                             "C5": time_config("ssd512_vgg16", 32, 0, args.steps, args.warmup, dev)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.model, g, 1002)
    if distributed:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner to the C stdout of rank 0; push it out first so that the JSON line is the last line
        C.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(result), flush=True)
    return result


if __name__ == "__main__":
    main()
