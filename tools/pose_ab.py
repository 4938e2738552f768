import os,sys,json
sys.path.insert(0,os.getcwd())
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api
from tools.run_configs import v1_frames, pad_batch, ev_time
ctx=gmmloc_amd.Context(0)
for (seq,cap,lo,hi) in (("V1_03_difficult",100000,150,1200),("V1_01_easy",100000,300,300)):
    mean,cov,cam,frames=v1_frames(seq,cap,lo,hi,3)
    prm=api.Params()
    pose,Xw,obs,octv=pad_batch(torch,frames,hi)
    t=ev_time(torch,lambda: gmmloc_amd.optimize_current_pose(ctx,cam,prm,pose.clone(),Xw,obs,octv),3,ctx.stream)
    big=[torch.cat([x]*8) for x in (pose,Xw,obs,octv)]
    t8=ev_time(torch,lambda: gmmloc_amd.optimize_current_pose(ctx,cam,prm,big[0].clone(),big[1],big[2],big[3]),2,ctx.stream)
    print(os.environ.get("GMMLOC_HIP_LIB","default").split("/")[-1], os.environ.get("GMMLOC_POSE_WAVES"), seq, len(frames), "frames/s", round(len(frames)/t), " x8 batch:", round(8*len(frames)/t8))
