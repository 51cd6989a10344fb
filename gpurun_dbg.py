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
for maxc in (200,37,5):
    a=g.good_features(maxc)
    print(maxc,len(a),np.array_equal(a,ref[:maxc]), flush=True)
