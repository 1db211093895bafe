import sys, json, os, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, oracle
from conftest import GOLDEN, load_decode_case
from nvimagecodec_amd.lowlevel import BatchDecoder
import bench
from nvimagecodec_amd.synth import synth_image
M=json.load(open(os.path.join(GOLDEN,'manifest.json')))
dec=BatchDecoder(0,8)
cases=[(e,load_decode_case(e)) for e in M['decode'] if e['progressive']]
jp=[c[1][0] for c in cases]
outs,st=dec.decode(jp, fmt='rgb', gpu_huffman=True, check=False)
torch.cuda.synchronize()
print('gpu decoded', dec.stats()['gpu_entropy_images'], 'of', len(jp), 'statuses', set(st))
bad=0
for (e,(j,rgb)),o in zip(cases,outs):
    ref = rgb if rgb is not None else oracle.decode(j)
    if not np.array_equal(o.cpu().numpy(), ref):
        bad+=1; print('MISMATCH', e['name'])
print('bad', bad)
srcs=[bench._pil_encode(synth_image(1920,1080,seed=900+k),90,'444',progressive=True) for k in range(4)]
refs=[torch.from_numpy(np.ascontiguousarray(oracle.decode(j).transpose(2,0,1))).cuda() for j in srcs]
batch=[srcs[i%4] for i in range(128)]
outs=dec.allocate_outputs(batch,'rgb_planar')
for rep in range(3):
    t0=time.perf_counter()
    dec.decode(batch, fmt='rgb_planar', outs=outs, gpu_huffman=True)
    torch.cuda.synchronize()
    t=time.perf_counter()-t0
    print('batch128 1080p 444 prog: %.1f ms -> %.0f img/s' % (t*1e3, 128/t), 'gpu images', dec.stats()['gpu_entropy_images'])
print('parity', all(torch.equal(o, refs[i%4]) for i,o in enumerate(outs)))
