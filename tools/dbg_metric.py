import sys; sys.path.insert(0,'.')
from multimesh_amd import synth
from multimesh_amd.device import Context
pa,ca=synth.hex_mesh(216,seed=1); pb,_=synth.hex_mesh(216,seed=7)
ctx=Context(0)
f=synth.vector_field(pa)[:1]
v,nf=ctx.interpolate_hex8(pa,ca,pb,f)
print("nf",nf)
