import sys, torch, numpy as np
sys.path.insert(0, ".")
from gftorf_amd import reference_network
from oracle import deform_ref
dev = torch.device("cuda:0")
params = deform_ref.random_params(3)
net = reference_network(); net.load_state_dict({k: torch.tensor(v) for k, v in params.items()}); net = net.to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
x = torch.rand((n, 3), device=dev); t = torch.full((1, 1), 0.4, device=dev).expand(n, -1)
g1, g2 = torch.randn((n, 3), device=dev), torch.randn((n, 16, 3), device=dev)
for _ in range(4):
    a, _, b, _ = net(x, t)
    torch.autograd.backward([a, b], [g1, g2])
    net.zero_grad(set_to_none=True)
torch.cuda.synchronize()
