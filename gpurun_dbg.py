import sys; sys.path.insert(0,'.')
import numpy as np
import larvio_amd
from larvio_amd import ops
from oracle import lvo
from tests.conftest import synth_frames
img=synth_frames(70,2)[0][1]
ctx=larvio_amd.Context()
g=ops.Pyramid(ctx,752,480).build(img,clahe=True)
o=lvo.LkPyramid(lvo.clahe(img))
ref=o.good_features(200)
for maxc in (200,37,37,100,5,1,64,65,200):
    a=g.good_features(maxc)
    ok=np.array_equal(a,ref[:maxc])
    print(maxc,len(a),ok, a[:4].tolist())
