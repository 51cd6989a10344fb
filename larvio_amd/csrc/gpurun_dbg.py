g2=ops.Pyramid(ctx,752,480).build(img,clahe=True)
for maxc in (37,200,37,37,200,13):
    a=g2.good_features(maxc)
    print('g2',maxc,len(a),np.array_equal(a,ref[:maxc]))
