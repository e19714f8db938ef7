import ctypes as C, numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from hawq_amd import _lib
from hawq_amd.packing import pack_conv_weight, pack_ctab
from hawq_amd.quant_utils import requant_table
lib=_lib
lib.load()
rng=np.random.default_rng(0)
for (n,h,w,cin,cout,tile) in [(128,14,14,256,256,-1),(128,14,14,256,256,0),(128,7,7,512,512,-1),(128,7,7,512,512,0),(128,56,56,64,64,-2),(128,28,28,128,128,-1),(128,28,28,128,128,0)]:
    x=torch.from_numpy(rng.integers(0,128,(n,h,w,cin)).astype(np.int8)).cuda()
    wt=rng.integers(-127,128,(cout,cin,3,3)).astype(np.int64)
    b=rng.integers(-2000,2000,cout).astype(np.int64)
    r=torch.from_numpy((rng.uniform(2e-5,3e-4,cout)*0.7).astype(np.float32))
    m,e=requant_table(torch.ones(1),r,torch.tensor([0.7]))
    wd=torch.from_numpy(pack_conv_weight(wt,8)).cuda(); bd=torch.from_numpy(b.astype(np.int32)).cuda()
    ct=torch.from_numpy(pack_ctab(b,m,e)).cuda(); md=torch.from_numpy(m).cuda(); ed=torch.from_numpy(e).cuda()
    out=torch.zeros(n*h*w*cout,dtype=torch.uint8,device='cuda')
    a=lib.ConvArgs()
    a.in_,a.wgt,a.bias=x.data_ptr(),wd.data_ptr(),bd.data_ptr()
    a.N,a.H,a.W,a.Cin,a.Cout,a.KH,a.KW,a.stride,a.pad=n,h,w,cin,cout,3,3,1,1
    a.in_bits=a.w_bits=8; a.tile=lib.load().hawq_conv2d_num_tiles()+tile; a.epilogue=1; a.relu=1; a.m,a.e,a.ctab=md.data_ptr(),ed.data_ptr(),ct.data_ptr(); a.fast_tables=1
    a.out_q,a.out_bits,a.q_lo,a.q_hi=out.data_ptr(),8,-128,127
    for _ in range(2): lib.call("hawq_conv2d", C.byref(a), None)
    torch.cuda.synchronize()
