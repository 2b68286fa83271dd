import sys, time, json
sys.path.insert(0,'/root/repo')
import torch, wxpkg
pkg=wxpkg.load_package()
from weather_sandbox_amd import devtools
X,Y=16384,2048
gui=pkg.params.merge_settings(None); gui["sunAngle"]=50.0
u=pkg.params.uniforms_from_gui(gui,Y,quad_scale=0); u["enablePrecipitation"]=0
for sigma in (0.2,0.3,0.4):
    h=pkg.engine.Handle(X,Y,0)
    h.setup_columns(pkg.synth.terrain_columns(X,Y)); h.set_params(pkg.params.fill_struct(pkg.params.WxParams(),u),u["initial_T"])
    devtools.seed_flow(h,sigma)
    out=[]
    for k in range(6):
        h.profile(True); h.sync(); t=time.perf_counter(); h.step(40); h.sync(); dt=time.perf_counter()-t
        pr=h.profile_read(); h.profile(False)
        fs=devtools.flow_stats(h)
        out.append((round(dt/40*1e3,3), round(fs["rms_v"],3), round(fs["max_v"],3), fs["cells_component_ge_0.9"]))
    print(sigma,out,flush=True)
    h.close()
