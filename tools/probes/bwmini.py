import os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qcc_amd import device, gates, native
n=30
st=device.DeviceState(n,128)
st.init_basis(0); h=gates.hadamard()
for q in range(n): st.apply1(h,q)
st.sync()
def timed(fn,reps=5):
    st.sync(); st.reset_stats(); st.timer_begin()
    for _ in range(reps): fn()
    ms=st.timer_end(); s=st.stats()
    return round(s['bytes_algorithmic']/reps/(ms/reps*1e-3)/1e9)
out={'U':os.environ.get('QH_GATE_U'),'NT':os.environ.get('QH_GATE_NT')}
out['H']={b:timed(lambda: st.apply1(h,n-1-b)) for b in (0,3,8,12,20,29)}
out['T']={b:timed(lambda: st.apply1(gates.tgate(),n-1-b)) for b in (3,8,20,29)}
out['cu1']={f'{c},{t}':timed(lambda: st.applyc(gates.u1(0.3),n-1-c,n-1-t)) for c,t in ((3,4),(6,7),(10,29),(20,21),(3,20))}
print(json.dumps(out))
