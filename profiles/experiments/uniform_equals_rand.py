import torch
dev=torch.device("cuda:0")
for seed in (1,77,5000):
    torch.manual_seed(seed); a=torch.rand((7,240,320),device=dev)*2-1
    torch.manual_seed(seed); b=torch.empty((7,240,320),device=dev).uniform_(-1.0,1.0)
    torch.manual_seed(seed); c=torch.zeros((7,240,320),device=dev); c.uniform_(-1.0,1.0)
    print(seed, torch.equal(a,b), torch.equal(a,c), float((a-b).abs().max()))
