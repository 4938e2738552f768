import sys; sys.path.insert(0,'.')
import torch, gmmloc_amd
from gmmloc_amd import _lib
_lib.load()
print(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l))
print(torch.cuda.current_stream().cuda_stream)
