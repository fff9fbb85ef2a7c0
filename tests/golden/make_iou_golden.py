"""Golden vectors for the silhouette IoU (SURVEY f-4), produced by the reference's OWN function definitions.

The experiment scripts cannot be imported (they import the CUDA extensions and run on import), so the definitions
of `iou_loss` (opt_shape.py) and `iou`, `iou_loss`, `multiview_iou_loss` (train_reconstruction.py) are located with
`ast` in /root/reference at generation time and executed in an empty namespace with torch.  Only inputs and outputs are
stored.  No-op when /root/reference is absent.

    python tests/golden/make_iou_golden.py
"""
import ast
import os

import numpy as np
import torch

REF = '/root/reference/experiments'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'iou')


def functions(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {'torch': torch}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, 'exec'), ns)
    return ns


def main():
    if not os.path.isdir(REF):
        print('reference not present; nothing to do')
        return
    os.makedirs(OUT, exist_ok=True)
    a = functions(os.path.join(REF, 'opt_shape.py'), {'iou_loss'})
    b = functions(os.path.join(REF, 'train_reconstruction.py'), {'iou', 'iou_loss', 'multiview_iou_loss'})
    g = torch.Generator().manual_seed(11)
    pred = torch.rand(5, 24, 24, generator=g) ** 2
    pred[1] = 0                                                # empty prediction
    target = (torch.rand(5, 24, 24, generator=g) > 0.6).float()
    target[3] = 0                                              # empty target
    views = [torch.rand(3, 4, 16, 16, generator=g) for _ in range(4)]
    ta, tb = torch.rand(3, 4, 16, 16, generator=g), torch.rand(3, 4, 16, 16, generator=g)
    out = dict(pred=pred.numpy(), target=target.numpy(),
               iou_loss_opt_shape=a['iou_loss'](pred, target).numpy(),
               iou_loss_train_reconstruction=b['iou_loss'](pred, target).numpy(),
               views=np.stack([v.numpy() for v in views]), targets_a=ta.numpy(), targets_b=tb.numpy(),
               multiview_iou_loss=b['multiview_iou_loss'](views, ta, tb).numpy())
    np.savez_compressed(os.path.join(OUT, 'iou.npz'), **out)
    print('wrote', os.path.join(OUT, 'iou.npz'), {k: (float(v) if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})


if __name__ == '__main__':
    main()
