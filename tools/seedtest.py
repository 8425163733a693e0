import sys, torch
sys.path.insert(0, '.')
from gcanet_amd import dgcnn
from bench import synth_clouds
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype='bf16').to(dev)
pts, nrm = synth_clouds(range(2), 8192, dev)
with torch.no_grad():
    x = torch.cat([pts, nrm], -1)
    xcm = x.transpose(1, 2).contiguous()
    idx1 = dgcnn.knn_points_normals(xcm, 64, 64)
    x1, x1cm = dgcnn.edge_conv_pm(x, idx1, m.encoder.conv1._modules['0'].weight, m.encoder.bn1, 'bf16')
    idx2 = dgcnn.knn(x1cm, 64, 64)
    x2, x2cm = dgcnn.edge_conv_pm(x1, idx2, m.encoder.conv2._modules['0'].weight, m.encoder.bn2, 'bf16')
    idx3 = dgcnn.knn(x2cm, 64, 64)
    for name, feat, seed, true in (("layer2", x1, idx1, idx2), ("layer3", x2, idx2, idx3)):
        q = torch.arange(0, 8192, 37, device=dev)
        f = feat[0]
        d = torch.cdist(f[q], f) ** 2                      # (Q, N)
        dseed = torch.gather(d, 1, seed[0, q])
        thr = dseed.max(1, keepdim=True)[0]
        npass = (d < thr).sum(1).float()
        overlap = (seed[0, q].unsqueeze(2) == true[0, q].unsqueeze(1)).any(2).float().sum(1)
        print(name, "candidates below seed threshold: mean %.0f median %.0f max %.0f | overlap with true kNN: %.1f / 64" % (
            npass.mean(), npass.median(), npass.max(), overlap.mean()))
