import numpy as np, collections, sys
sys.path.insert(0,'/root/repo')
from gym_rem2d_amd import synthetic, make_terrain, Morphology
from oracle import oracle as O
O.build()
specs=[s for s in synthetic.lsystem_specs(range(400)) if s.n_bodies>=12][:48]
print(len(specs),'heavy creatures')
t=make_terrain(4, flat=True); xs,ys,polys=t.f32(); ot=O.Terrain(xs,ys,None,t.friction)
m=Morphology.from_specs(specs,16)
jr=m.arrays['jround'].reshape(m.n_envs,16)
par=m.arrays['parent'].reshape(m.n_envs,16)
stats=collections.Counter(); phases_hist=collections.Counter(); frac=[]
worlds=[O.World.from_morph(ot,m.as_dict(),e,flags=O.FLAG_CONTINUOUS) for e in range(m.n_envs)]
per_creature_phase_sets=[]
for step in range(400):
    for w in worlds: w.env_step()
    if step>=100 and step%20==0:
        snap=[]
        for e,w in enumerate(worlds):
            J=w.joints()  # [nj][6]: ... limitState at [5]
            nj=w.n_joints
            P=max(1,int((jr[e]>>16).max()&0xff))
            rounds=[int(jr[e][k+1]&0xff) for k in range(nj)]
            lim=[int(J[k][5])!=0 for k in range(nj)]
            frac.append(np.mean(lim))
            ph=set(r%P for r,l in zip(rounds,lim) if l)
            snap.append((P,ph,[r%P for r in rounds]))
            phases_hist[(P,len(ph))]+=1
        per_creature_phase_sets.append(snap)
print('fraction of joints at a limit: mean %.3f'%np.mean(frac))
print('(P, #phases with a limit-active joint) histogram:', sorted(phases_hist.items()))
# tiles of 4 consecutive creatures: expensive slots per iteration with rotation 0 vs best rotation
import itertools
tot0=tot1=totslots=0
for snap in per_creature_phase_sets:
    for i in range(0,len(snap)-3,4):
        tile=snap[i:i+4]; P=max(p for p,_,_ in tile)
        def cost(rots):
            exp=set()
            for (p,ph,_),r in zip(tile,rots):
                for x in ph: exp.add((x+r)%P)
            return len(exp)
        c0=cost((0,0,0,0))
        best=min(cost(r) for r in itertools.product(range(P),repeat=4))
        tot0+=c0; tot1+=best; totslots+=P
print('expensive joint slots per iteration: unrotated %.3f of slots, best rotation %.3f of slots'%(tot0/totslots, tot1/totslots))
