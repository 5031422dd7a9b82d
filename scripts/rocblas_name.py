import torch
X = torch.randn(200000, 2000, device="cuda"); W = torch.randn(2000, 512, device="cuda"); D = torch.randn(200000, 512, device="cuda")
for _ in range(2):
    torch.mm(X, W); torch.mm(X.t(), D)
torch.cuda.synchronize()
