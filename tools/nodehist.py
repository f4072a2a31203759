import sys, numpy as np, torch
sys.path.insert(0,'.')
import bench
from morb_slam_amd import ORBextractor, ORBmatcher
from morb_slam_amd.synth import make_vocabulary
B=8
frames=torch.from_numpy(bench.make_batch(list(range(B)),B,seed=0)).cuda()
images=frames.view(2*B,bench.H,bench.W)
ext=ORBextractor(1200,1.2,8,20,7)
m=ORBmatcher(0.7,True)
kps,desc,cnt,_=ext.extract_batch(images)
vd,vf=make_vocabulary(10,6,seed=0)
w,n=m.bow_transform(desc,cnt,torch.from_numpy(vd).cuda(),torch.from_numpy(vf).cuda(),10,6,4)
n0=n[0,:int(cnt[0])].cpu().numpy()
u,c=np.unique(n0,return_counts=True)
print("nodes",len(u),"max",c.max(),"sorted top",np.sort(c)[::-1][:15],"median",np.median(c))
