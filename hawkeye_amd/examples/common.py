import torch


def warmup_cosine(optimizer, config):
    """Linear warm-up then cosine annealing, stepped per epoch (Examples/CBCNN.py:35-45, MPN.py:20-30)."""
    main = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=config['T_max'] - config['warmup_epochs'])
    warm = torch.optim.lr_scheduler.LinearLR(optimizer, start_factor=config['lr_warmup_decay'],
                                             total_iters=config['warmup_epochs'])
    return torch.optim.lr_scheduler.SequentialLR(optimizer, schedulers=[warm, main],
                                                 milestones=[config['warmup_epochs']])
