import sys, torch, numpy as np
sys.path.insert(0, "mc-cnn-python_amd/src"); sys.path.insert(0, "tests")
import stereo_device as sd, synthetic
H,W,D=500,750,256
L,R,_,_,_=synthetic.make_pair(H,W,D,seed=100)
dl=torch.from_numpy(L[:,:,0]).cuda()
sup=sd.cross_arms(dl,0.02,14)
c=sd.support_count(sup).float()
a=sd.support_arms(sup).float()
print("region size mean %.1f max %d ; arms mean"%(c.mean().item(), int(c.max())), a.mean(dim=(0,1)).tolist())
# wave-max: 64 consecutive pixels
cm=c[:, :704].reshape(H,11,64).max(dim=2).values
print("mean of wave-max region size %.1f"%cm.mean().item())
