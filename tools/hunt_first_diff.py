"""dev tool: forwards in flight vs one forward at a time, tensor by tensor (DN_WS_REUSE=0 keeps every intermediate): prints the FIRST tensors that
differ, and for pooled partial sums which side agrees with the depthwise output they were summed from.   usage: hunt_first_diff.py <batch> <depth>"""
import ctypes as C, os, sys
os.environ["DN_WS_REUSE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demonet_amd import _lib, models, synth
from demonet_amd.pipeline import ForwardPipeline
L = _lib.lib()
n, depth = int(sys.argv[1]), int(sys.argv[2])
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
g = m.graph
W, H = g.size
batches = [torch.from_numpy(synth.images(40 + i, n, H, W)).cuda() for i in range(6)]
def snap(handle, ws):
    out = {}
    for tid in range(len(g.tensors)):
        p, sz = C.c_void_p(), C.c_size_t()
        rc = L.dn_tensor_ptr(C.c_void_p(handle), C.c_void_p(ws.data_ptr()), n, tid, C.byref(p), C.byref(sz))
        if rc != 0 or not sz.value:
            continue
        off = p.value - ws.data_ptr()
        out[tid] = ws[off:off + sz.value].clone()
    return out
with ForwardPipeline(m, n, depth=depth) as pipe:
    h = m._handle
    ref = []
    for b in batches:
        r = [t.clone() for t in m.forward_batch(b)]
        torch.cuda.synchronize()
        key = next(k for k in m._bufs if k[0] == n)
        ref.append((r, snap(h, m._bufs[key]["ws"])))
    found = False
    for rnd in range(6):
        base = pipe.n
        for k, b in enumerate(batches):
            pipe.submit(b)
            j = k - (depth - 1)
            if j < 0:
                continue
            out = [x.clone() for x in pipe.result(base + j)]
            if all(torch.equal(a, c) for a, c in zip(ref[j][0], out)):
                continue
            s = snap(h, pipe.slots[(base + j) % depth].ws)
            producers = {nd.out: (i, nd.op, nd.conv_key or nd.fc1_key or "") for i, nd in enumerate(g.nodes)}
            bad = [tid for tid in sorted(s) if tid in ref[j][1] and not torch.equal(s[tid], ref[j][1][tid])]
            print(f"round {rnd} batch {j}: outputs differ; differing tensors: {len(bad)} of {len(s)}")
            for tid in bad[:6]:
                a16 = s[tid].view(torch.float16) if g.tensors[tid].kind == 'act' else s[tid].view(torch.float32)
                b16 = ref[j][1][tid].view(a16.dtype)
                d = (a16.float() - b16.float()).abs()
                per_img = d.view(n, -1).amax(1)
                print("   tensor", tid, "produced by op", producers.get(tid), "max|d|", float(d.max()), "images", [int(i) for i in torch.nonzero(per_img > 0).flatten()[:8]])
                if g.tensors[tid].kind == 'pool':
                    t = g.tensors[tid]
                    rows = a16.numel() // (n * t.c)
                    A = a16.view(n, rows, t.c); B = b16.view(n, rows, t.c)
                    nz = torch.nonzero((A != B).any(2))
                    print("      pool tensor c =", t.c, "rows per image =", rows, "differing (image, row):", nz.tolist()[:12])
                    # which side is right? the dw output tensor (tid - 1, fp16-rounded, ReLU/hswish applied) summed over the image vs the summed partial rows
                    if (tid - 1) in s and g.tensors[tid - 1].kind == 'act' and g.tensors[tid - 1].c == t.c:
                        to = g.tensors[tid - 1]
                        outp = s[tid - 1].view(torch.float16).view(n, to.h * to.w, to.c).float().sum(1)
                        for (ii, rr) in nz.tolist()[:2]:
                            ch = torch.nonzero(A[ii, rr] != B[ii, rr]).flatten().tolist()
                            print("      image", ii, "row", rr, "channels", ch)
                            print("      sum of rows pipelined", [round(float(A[ii, :, c].sum()), 3) for c in ch[:8]])
                            print("      sum of rows serial   ", [round(float(B[ii, :, c].sum()), 3) for c in ch[:8]])
                            print("      sum of dw output     ", [round(float(outp[ii, c]), 3) for c in ch[:8]])
                            print("      row value pipelined / serial", [(round(float(A[ii, rr, c]), 3), round(float(B[ii, rr, c]), 3)) for c in ch[:8]])
                            print("      differing channels:", len(ch), "of", t.c, " components (c % 8):", sorted(set(c % 8 for c in ch)), " max |d|", float((A[ii, rr] - B[ii, rr]).abs().max()))
                            # is the pipelined row STALE, i.e. equal to what another batch leaves at this position (the slot's previous forward)?
                            for jb in range(len(ref)):
                                if jb != j and tid in ref[jb][1]:
                                    O = ref[jb][1][tid].view(a16.dtype).view(n, rows, t.c)
                                    same = int((A[ii, rr] == O[ii, rr]).sum())
                                    samech = int((A[ii, rr][ch] == O[ii, rr][ch]).sum())
                                    print(f"         vs serial batch {jb}: {same} of {t.c} channels equal, {samech} of the {len(ch)} differing ones")
                    for (ii, rr) in nz.tolist()[:0]:
                        print("      pipelined", A[ii, rr, :6].tolist(), "\n      serial   ", B[ii, rr, :6].tolist(), " channels differing", int((A[ii, rr] != B[ii, rr]).sum()))
            found = True
            break
        if found:
            break
    print("found" if found else "no mismatch in 6 rounds")
