import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
mode = sys.argv[1]
if mode == 'torch_first':
    import torch
    print('torch sees', torch.cuda.is_available(), torch.cuda.device_count(), torch.version.hip)
    from qcc_amd import native, device
    print('engine sees', native.device_count())
    t = torch.zeros(2 << 10, dtype=torch.float64, device='cuda')
    st = device.DeviceState(10, 128, device_ptr=t.data_ptr())
    st.init_basis(3); 
    from qcc_amd import gates
    st.apply1(gates.hadamard(), 0); st.sync()
    torch.cuda.synchronize()
    print('norm2', st.norm2(), 'torch view', float((t*t).sum()))
else:
    from qcc_amd import native, device
    print('engine sees', native.device_count())
    import torch
    print('torch sees', torch.cuda.is_available(), torch.cuda.device_count())
import subprocess
print(subprocess.run("grep -E 'libamdhip64|libhsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid(), shell=True, capture_output=True, text=True).stdout)
