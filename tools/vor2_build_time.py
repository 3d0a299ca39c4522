"""Host time of the candidate-table build (pqhip_vor2_tables_host through ra.vor2_tables); no GPU needed."""
import numpy as np, time, reductive_amd as ra
rng=np.random.default_rng(1)
for M,K,ds in [(150,256,2),(10,128,2),(128,256,1)]:
    q=rng.standard_normal((M,K,ds),dtype=np.float32)
    t=time.time(); w,off=ra.vor2_tables(q); dt=time.time()-t
    print(M,K,ds,'%.0f ms'%(dt*1e3))
