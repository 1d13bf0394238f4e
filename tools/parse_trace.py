import numpy as np, sys
raw=np.load(sys.argv[1]); n=int(sys.argv[2])
raw=raw.reshape(-1)
a=raw[:n*4].reshape(n,4); q=raw[n*4:n*4+n*8].reshape(n,8)
dur=(a[:,1]-a[:,0])/100.0
whole=q[:,7].astype(float); sm=q[:,5].astype(float); src=q[:,6].astype(float)
ok=whole>0
print('items',n,'wave us mean %.2f'%dur.mean(),'whole cycles mean %.0f (%.2f GHz)'%(whole[ok].mean(), (whole[ok]/dur[ok]/1e3).mean()))
print('smoothness pass cycles mean %.0f  source passes (both) %.0f  start-up + write-out %.0f'%(sm[ok].mean(), src[ok].mean(), (whole-sm-src)[ok].mean()))
