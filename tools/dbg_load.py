import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
def maps():
    libs = set()
    for l in open('/proc/self/maps'):
        if 'hip64' in l or 'hsa-runtime' in l:
            libs.add(l.split()[-1])
    return sorted(libs)
if order == 'gfo_first':
    import gf_orb_slam2_amd as G
    e = G.ORBextractor(500, 1.2, 8, 20, 7)
    print('after gfo', maps())
    import torch
    print('after torch import', maps())
    try:
        torch.cuda.init(); print('torch ok', torch.cuda.device_count())
    except Exception as ex: print('torch FAIL', ex)
else:
    import torch
    torch.cuda.init(); print('torch ok', maps())
    import gf_orb_slam2_amd as G
    e = G.ORBextractor(500, 1.2, 8, 20, 7)
    print('gfo ok', maps())
