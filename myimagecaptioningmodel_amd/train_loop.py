"""The step loop of the reference's train() (/root/reference/ImageCaptioning/train.py:133-149,172)
over this package's engine: per step feed -> train step -> NaN assertion -> running loss; per
epoch save_persistables into `<checkpoint_path>/checkpoint` and the logger's resume JSON."""
import os

import numpy as np

from . import ckpt


def log(log_path, content, end='\n'):
    """tools/logger.py:83-88: print and append to <log_path>/log.txt."""
    print(content, end=end)
    with open(os.path.join(log_path, 'log.txt'), 'a') as f:
        f.write(content + end)


def load_model(engine, conf, checkpoint_path, log_path, pretrained_encoder_path=None):
    """train.py:94-107.  First init (epoch == 1): optionally the pretrained encoder variables whose files exist
    (`PretrainedMobileNetPath`).  Resume: load_persistables from `<checkpoint_path>/checkpoint`; if the
    encoder_trainable flag differs from the one the run was started with, record the new one and -- when the encoder
    just became trainable -- reload the pretrained encoder (:103-107)."""
    if conf['epoch'] == 1:                                      # logger.is_first_init
        if pretrained_encoder_path is not None:
            ckpt.load_vars_existing(engine, pretrained_encoder_path)
        return
    ckpt.load_persistables(engine, os.path.join(checkpoint_path, 'checkpoint'))
    trainable = bool(engine.cfg['encoder_trainable'])
    if conf.get('train_encoder') != trainable:
        conf['train_encoder'] = trainable
        ckpt.save_resume_state(log_path, conf)
        if trainable and pretrained_encoder_path is not None:
            ckpt.load_vars_existing(engine, pretrained_encoder_path)


def train(engine, batches_per_epoch, max_epoch, checkpoint_path, log_path, log_every_n_step=150,
          pretrained_encoder_path=None, checkpoint_backup_every_n_epoch=0, export_params=False):
    """batches_per_epoch: callable epoch -> iterable of {'image': ..., 'caption': ...} feeds
    (the reader contract of reader.py:45-47,65).  Resumes from `<log_path>/config` like the reference: the epoch is
    written at the START of each epoch (train.py:134), so a crash inside epoch N restarts epoch N from the
    checkpoint written at the end of epoch N-1 (and a crash inside epoch 1 starts from scratch)."""
    conf = ckpt.load_resume_state(log_path, engine.cfg['encoder_trainable'])
    load_model(engine, conf, checkpoint_path, log_path, pretrained_encoder_path)
    for epoch in range(conf['epoch'], max_epoch + 1):
        conf['epoch'] = epoch                                   # written at the START of the epoch (train.py:134)
        ckpt.save_resume_state(log_path, conf)
        log(log_path, 'Epoch {}'.format(epoch))
        epoch_loss, step = 0.0, -1
        for step, data in enumerate(batches_per_epoch(epoch)):
            loss, lr = engine.train_step(data['image'], data['caption'])
            step_loss = loss.detach().cpu().numpy()
            engine.check_sync()
            if np.isnan(step_loss).any():                       # train.py:140-141
                raise AssertionError('Epoch:{} Step:{} Loss为Nan'.format(epoch, step + 1))
            epoch_loss += float(step_loss[0])
            if (step + 1) % log_every_n_step == 0:
                log(log_path, ' ' * 4 + 'Step {} Mean loss: {:6f} Step loss: {:6f}, lr: {}'.format(
                    step + 1, epoch_loss / (step + 1), float(step_loss[0]), str(np.float32(lr))))
        log(log_path, 'Epoch loss: {:7f}'.format(epoch_loss / max(1, step + 1)))
        ckpt.save_persistables(engine, os.path.join(checkpoint_path, 'checkpoint'))   # train.py:172 -> :73
        n = checkpoint_backup_every_n_epoch
        if n and epoch % n == 0:                                                      # :74-76
            ckpt.save_persistables(engine, os.path.join(checkpoint_path, 'checkpoint{}'.format(epoch)))
        if export_params:                                                             # :78-79
            ckpt.save_params(engine, os.path.join(checkpoint_path, 'params'))
    return conf
