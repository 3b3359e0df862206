"""Tester runtime (mirrors the reference's test.py:14-142): build the registered
model, strict-load `model.load`, run the validation split, report top-1."""
import logging
import os

import torch

from .config import setup_config
from .model.registry import MODEL
from .utils import AverageMeter, TqdmHandler, accuracy


class Tester:
    def __init__(self, config=None):
        self.config = config if config is not None else setup_config()
        cfg = self.config
        self.logger = logging.getLogger()
        self.logger.handlers = [TqdmHandler()]
        self.logger.setLevel(logging.INFO)
        self.device = self.select_device(cfg)
        self.transformer = self.get_transformer(cfg.dataset.transformer)
        self.collate_fn = self.get_collate_fn()
        self.dataset = self.get_dataset(cfg.dataset)
        self.dataloader = self.get_dataloader(cfg.dataset)
        self.model = self.to_device(self.get_model(cfg.model), parallel=True)
        self.average_meters = {'acc': AverageMeter()}

    def select_device(self, cfg):
        """This process's GPU; no CPU path (the HIP heads have none - the CPU reference is test infrastructure)."""
        if not (isinstance(cfg.experiment.cuda, list) and cfg.experiment.cuda and torch.cuda.is_available()):
            raise RuntimeError('hawkeye_amd evaluates on MI355X only: set experiment.cuda: [0] and run on a GPU host')
        device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
        torch.cuda.set_device(device)           # the C ABI launches on the CURRENT device's stream (functional.stream())
        return device

    def get_transformer(self, config):
        from . import transforms
        dev = bool(config['device_finalize']) if 'device_finalize' in config else False
        return transforms.ClassificationPresetEval(crop_size=config['image_size'], resize_size=config['resize_size'],
                                                   device_finalize=dev)

    def get_collate_fn(self):
        return None

    def get_dataset(self, config):
        from . import data
        if config.name == 'synthetic':
            return data.SyntheticDataset(config.samples if 'samples' in config else 256,
                                         config.transformer.image_size, self.config.model.num_classes, seed=1)
        return data.FGDataset(config.root_dir, os.path.join(config.meta_dir, 'val.txt'), self.transformer)

    def get_dataloader(self, config):
        from torch.utils.data import DataLoader
        return DataLoader(self.dataset, config.batch_size, num_workers=config.num_workers, pin_memory=True,
                          shuffle=False, collate_fn=self.collate_fn)

    def get_model(self, config):
        model = MODEL.get(config.name)(config)
        assert 'load' in config and config.load != '', 'There is no valid `load` in config[model.load]!'   # test.py:69
        model.load_state_dict(torch.load(config.load, map_location='cpu'))
        return model

    def to_device(self, m, parallel=False):
        if isinstance(m, dict) and 'u8' in m:              # uint8 crops from the workers (transformer.device_finalize)
            from . import functional as HF
            return HF.image_finalize(m['u8'].to(self.device, non_blocking=True), m['erase'].to(self.device, non_blocking=True))
        return m.to(self.device, non_blocking=True) if isinstance(m, torch.Tensor) else m.to(self.device)

    def test(self):
        self.logger.info(f'Testing model from {self.config.model.load}')
        self.validate()
        self.report()

    def validate(self):
        self.model.train(False)
        with torch.no_grad():
            for data in self.dataloader:
                self.batch_validate(data)

    def batch_validate(self, data):
        images, labels = self.to_device(data['img']), self.to_device(data['label'])
        out = self.model(images)
        out = out[0] if isinstance(out, (tuple, list)) else out
        self.average_meters['acc'].update(accuracy(out, labels, 1), images.size(0))

    def report(self):
        self.logger.info(f"top-1 accuracy: {self.average_meters['acc'].avg:.3f}")


if __name__ == '__main__':
    Tester().test()
