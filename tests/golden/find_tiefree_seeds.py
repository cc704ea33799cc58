"""Authoring-time helper: scan image seeds with the ORACLE for post-process instances that are far from any\norder/threshold flip (see oracle.selection_margins). usage: find_tiefree_seeds.py <model> <num_classes> <lo> <hi>"""
import sys, numpy as np, torch
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'oracle'))
from demonet_amd import spec, synth
import ssd_oracle as so
torch.set_num_threads(8)
name=sys.argv[1]; ncls=int(sys.argv[2]); lo=int(sys.argv[3]); hi=int(sys.argv[4])
g = spec.GRAPHS[name](num_classes=ncls)
sd = synth.state_dict(g, 0)
post=dict(g.post)
if name=="ssd_lite_mobilenet_v2": post['score_thresh']=0.02
o = so.OracleSSD(name, sd, ncls, size=g.size, **post)
for s in range(lo,hi):
    img = torch.from_numpy(synth.images(s,1,g.size[1],g.size[0])[0])
    d, raw = o([img], return_intermediates=True)
    m = so.selection_margins(d[0]['softmax'], d[0]['decoded'], post['score_thresh'], post['nms_thresh'], post['topk_candidates'], post['detections_per_img'])
    ok = min(m['topk_gap'],m['order_gap'],m['final_gap'],m['thresh_gap'])>1e-6 and m['iou_gap']>5e-6
    print(s, "OK" if ok else "--", len(d[0]['scores']), {k: float('%.3g'%v) for k,v in m.items()}, flush=True)
