"""One-off long soak (not part of the test suite): N full-resolution cfg2 frames through the HIP path and the CPU oracle side by
side; every frame's flags, counters and feature set must match bit for bit.  usage: python tools/long_soak.py [frames] [step_m] [movers] [window] [float_sums]
Round-1 results on an MI355X box: 200 frames, step 0.08 m: 0 mismatching frames, 199 poses, max |dt| 1.1e-9 m, ATE 5.2 cm over
15.9 m vs ground truth, 5.5e-10 m vs the oracle; step 0.5 m (the camera flies out of the rendered scene after ~40 frames, so most
frames take the failure paths): 0 mismatching frames as well.
Round 2, with the independently moving layer (movers 0.3, step 0.08), final build: 200 frames, 0 mismatching frames, 198 poses,
max |dt| 1.8e-9 m, ATE 4.1 cm over 15.9 m vs ground truth, 8e-10 m vs the oracle."""
import sys, os, time, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/tests')
import oracle_lib as orc
from stereo_visual_odometry_amd import api, synthetic as syn
orc.set_threads(16)
cal=syn.KITTI00; NF=int(sys.argv[1]) if len(sys.argv)>1 else 200
t=time.time(); seq=syn.StereoSequence(cal=cal,n_frames=NF,seed=0x5EED0042,step=float(sys.argv[2]) if len(sys.argv)>2 else 0.5,cell_px=17.6,yaw_amp_deg=0.5,movers=float(sys.argv[3]) if len(sys.argv)>3 else 0.0); print('rendered',NF,'frames in',round(time.time()-t,1),'s',flush=True)
WIN=int(sys.argv[4]) if len(sys.argv)>4 else 21
FS=int(sys.argv[5]) if len(sys.argv)>5 else 0          # 1 = lk_float_sums on both sides
over=dict(win_w=WIN,win_h=WIN,max_level=3,ransac_iterations=100,max_translation_norm=2.0,lk_float_sums=FS)
Pl,Pr=syn.projection_matrices(cal)
o=orc.VisualOdometry(orc.default_config(**over)); o.initalize_projection_matricies(Pl,Pr)
g=api.VisualOdometry(cfg=api.default_config(**over)); g.initalize_projection_matricies(Pl,Pr)
bits=lambda a: np.ascontiguousarray(a,np.float32).view(np.uint32)
bad=0; nok=0; maxdt=0; maxdr=0; est=[]; ref=[]; gt=[]
for k in range(NF):
    ok_o,T_o=o.stereo_callback(seq.left[k],seq.right[k]); ok_g,T_g=g.stereo_callback(seq.left[k],seq.right[k])
    so={f[0]:getattr(o.stats,f[0]) for f in o.stats._fields_}; sg=g.stats.as_dict()
    fo,fg=o.features(),g.features()
    same = ok_o==ok_g and so==sg and np.array_equal(bits(fo[0]),bits(fg[0])) and np.array_equal(fo[1],fg[1]) and np.array_equal(fo[2],fg[2])
    if not same: bad+=1; print('MISMATCH frame',k,so,sg,flush=True)
    nok+=int(ok_g); maxdt=max(maxdt,np.abs(T_o[:3,3]-T_g[:3,3]).max()); maxdr=max(maxdr,np.abs(T_o[:3,:3]-T_g[:3,:3]).max())
    if k>0: est.append(T_g); ref.append(T_o); gt.append(seq.relative_motion(k))
    if k%50==0: print('frame',k,'ok',ok_g,'n_lk',sg['n_into_lk'],'inl',sg['n_inliers'],flush=True)
print('frames',NF,'mismatching',bad,'poses',nok,'max |dt|',maxdt,'max |dR|',maxdr)
print('ATE vs ground truth', syn.ate_rmse(syn.integrate(est),syn.integrate(gt)),'m over',sum(np.linalg.norm(x[:3,3]) for x in gt),'m ; vs oracle', syn.ate_rmse(syn.integrate(est),syn.integrate(ref)))
