"""Developer tool (GPU box): per-launch stage times on tiles from 4 K to 33 M paths of C4 — what a launch costs when it has
next to nothing to do (k_extend / k_shadow: 9 us at 4 096 paths)."""
import sys; sys.path.insert(0,'.')
import pbrs_amd
from pbrs_amd import scenes
sb,cfg=scenes.build_config("c4")
ctx=pbrs_amd.Context(0); ctx.upload(pbrs_amd.HostScene(sb))
for tile in ((0,0,64,64),(0,0,256,256),(0,0,1920,1080)):
    for spp in ((1,1),(4,4)):
        ctx.render(spp[0],spp[1],8,1,tile=tile)
        img,st=ctx.render(spp[0],spp[1],8,1,tile=tile,timing=True)
        n=tile[2]*tile[3]*spp[0]*spp[1]
        print(tile[2:],spp,"paths",n,{k:round(st[k]/max(1,st['launches_extend']),4) for k in ('ms_extend','ms_shade','ms_shadow')}, 'launches',st['launches_extend'], 'raygen',round(st['ms_raygen'],4),'acc',round(st['ms_accumulate'],4),'total',round(st['ms_total'],3))
