"""dev tool: is the depthwise kernel's result independent of what else runs on the chip?  Forwards in flight vs one forward at a time with
every pooling depthwise launch also dumping its per-thread fp32 outputs and pooled sums (dn_debug_dw_table). On the first forward whose
detections differ it classifies the first differing pooled tensor: per-thread outputs (conv accumulation), per-thread sums (epilogue),
or only the reduced row (LDS reduction / publish).
    usage: [DN_DW_ROWS=1] [DN_SE_SMALL=0] hunt_dw_rows.py <batch> <depth> <rounds>"""
import ctypes as C, os, sys
os.environ["DN_WS_REUSE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demonet_amd import _lib, models, synth
from demonet_amd.pipeline import ForwardPipeline
L = _lib.lib()
raw = C.CDLL(_lib.LIB_PATH)
n, depth, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
g = m.graph
W, H = g.size
NB = 6
batches = [torch.from_numpy(synth.images(40 + i, n, H, W)).cuda() for i in range(NB)]
pool_tids = [tid for tid, t in enumerate(g.tensors) if t.kind == "pool"]
prod = {nd.pool: nd for nd in g.nodes if getattr(nd, "pool", -1) >= 0}


def tptr(handle, ws, tid):
    p, sz = C.c_void_p(), C.c_size_t()
    rc = L.dn_tensor_ptr(C.c_void_p(handle), C.c_void_p(ws.data_ptr()), n, tid, C.byref(p), C.byref(sz))
    return (p.value, sz.value) if rc == 0 and sz.value else (None, 0)


def snap(handle, ws):
    out = {}
    for tid in range(len(g.tensors)):
        p, sz = tptr(handle, ws, tid)
        if p:
            off = p - ws.data_ptr()
            out[tid] = ws[off:off + sz].clone()
    return out


with ForwardPipeline(m, n, depth=depth) as pipe:
    h = m._handle
    m.forward_batch(batches[0])                      # creates the serial workspace
    torch.cuda.synchronize()
    key = next(k for k in m._bufs if k[0] == n)
    wss = [m._bufs[key]["ws"]] + [s.ws for s in pipe.slots]
    # debug table: (workspace, pool tensor) -> region
    REC = 40
    ptrs, owner = [], []
    max_rows = 0
    for wi, ws in enumerate(wss):
        for tid in pool_tids:
            p, sz = tptr(h, ws, tid)
            if p:
                ptrs.append(p); owner.append((wi, tid))
                max_rows = max(max_rows, sz // (4 * g.tensors[tid].c))
    stride = max_rows * 256 * REC * 4
    dbg = torch.zeros(len(ptrs) * stride, dtype=torch.uint8, device="cuda")
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    raw.dn_debug_dw_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    raw.dn_debug_dw_table(arr, len(ptrs), C.c_void_p(dbg.data_ptr()), C.c_size_t(stride))
    raw.dn_debug_clear_graphs.argtypes = [C.c_void_p]
    raw.dn_debug_clear_graphs(C.c_void_p(h))
    print(f"batch {n} depth {depth}: {len(pool_tids)} pooled tensors, {len(ptrs)} debug regions of {stride >> 20} MB; DN_DW_ROWS={os.environ.get('DN_DW_ROWS', '0')} DN_SE_SMALL={os.environ.get('DN_SE_SMALL', '1')}", flush=True)

    def region(wi, tid):
        i = owner.index((wi, tid))
        t = g.tensors[tid]
        nd = prod[tid]
        tw = 4 if nd.stride == 1 else 2
        rec = tw * 8 + 8
        p, sz = tptr(h, wss[wi], tid)
        rows = sz // (4 * t.c)          # n * nblocks
        return dbg[i * stride:i * stride + rows * 256 * rec * 4].view(torch.float32).view(rows, 256, rec).clone(), tw

    ref = []
    for b in batches:
        r = [t.clone() for t in m.forward_batch(b)]
        torch.cuda.synchronize()
        ref.append((r, snap(h, wss[0]), {tid: region(0, tid) for tid in pool_tids if (0, tid) in owner}))
    # serial repeatability of the dump itself
    r2 = [t.clone() for t in m.forward_batch(batches[-1])]
    torch.cuda.synchronize()
    print("serial repeat identical:", all(torch.equal(a, c) for a, c in zip(ref[-1][0], r2)), flush=True)
    found = 0
    checked = 0
    for rnd in range(rounds):
        base = pipe.n
        for k in range(NB + depth - 1):
            if k < NB:
                pipe.submit(batches[k])
            j = k - (depth - 1)
            if j < 0:
                continue
            if k >= NB:
                pipe.submit(batches[k % NB])        # keep the chip busy while the last tickets are read
            out = [x.clone() for x in pipe.result(base + j)]
            checked += 1
            if all(torch.equal(a, c) for a, c in zip(ref[j][0], out)):
                continue
            found += 1
            if found > 3:
                continue
            wi = 1 + (base + j) % depth
            pipe.drain()
            s = snap(h, wss[wi])
            bad = [tid for tid in sorted(s) if tid in ref[j][1] and not torch.equal(s[tid], ref[j][1][tid])]
            producers = {nd.out: (i, nd.op, nd.conv_key or nd.fc1_key or "") for i, nd in enumerate(g.nodes)}
            print(f"round {rnd} batch {j} slot {wi - 1}: outputs differ; differing tensors {len(bad)} of {len(s)}; first: {bad[:4]}", flush=True)
            for tid in bad[:2]:
                t = g.tensors[tid]
                if t.kind != "pool":
                    a16 = s[tid].view(torch.float16); b16 = ref[j][1][tid].view(torch.float16)
                    d = (a16.float() - b16.float()).abs()
                    print("   tensor", tid, "kind", t.kind, "producer", producers.get(tid), "max|d|", float(d.max()), "count", int((d > 0).sum()))
                    continue
                A = s[tid].view(torch.float32).view(-1, t.c); B = ref[j][1][tid].view(torch.float32).view(-1, t.c)
                nz = torch.nonzero((A != B).any(1)).flatten().tolist()
                print(f"   pool tensor {tid} c={t.c} producer k={prod[tid].k} s={prod[tid].stride}: differing rows {nz[:8]} of {A.shape[0]}")
                if (wi, tid) not in owner:
                    continue
                P, tw = region(wi, tid)
                Q, _ = ref[j][2][tid]
                for row in nz[:2]:
                    ch = torch.nonzero(A[row] != B[row]).flatten().tolist()
                    print(f"      row {row}: {len(ch)} channels differ, components {sorted(set(c % 8 for c in ch))}; first", [(c, float(A[row, c]), float(B[row, c])) for c in ch[:4]])
                    pa, pb = P[row], Q[row]                 # [256][rec]
                    dacc = torch.nonzero((pa[:, :tw * 8] != pb[:, :tw * 8]).any(1)).flatten().tolist()
                    dsum = torch.nonzero((pa[:, tw * 8:] != pb[:, tw * 8:]).any(1)).flatten().tolist()
                    print(f"      per-thread outputs differ in threads {dacc[:16]} ({len(dacc)}), per-thread sums differ in threads {dsum[:16]} ({len(dsum)})")
                    for th in (dacc or dsum)[:3]:
                        da = torch.nonzero(pa[th] != pb[th]).flatten().tolist()
                        print(f"         thread {th} (wave {th // 64} lane {th % 64}): record entries", [(e, float(pa[th, e]), float(pb[th, e])) for e in da[:8]])
                    c8 = t.c // 8
                    # does the row equal the fixed-order sum of the dumped per-thread sums?
                    for name, X, R in (("pipelined", pa, A[row]), ("serial", pb, B[row])):
                        tot = torch.zeros(t.c)
                        base_idx = (row % (A.shape[0] // n)) * 256
                        for th in range(256):
                            cg = (base_idx + th) % c8
                            tot[cg * 8:cg * 8 + 8] += X[th, tw * 8:].cpu()
                        print(f"         {name}: max |row - sum of dumped thread sums| = {float((tot - R.cpu()).abs().max()):.3e}")
    print(f"checked {checked} forwards in flight: {found} differ from their serial forward")
