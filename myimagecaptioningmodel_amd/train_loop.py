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


def dev_evaluation(engine, dev_batches, metric, log_path=None, index_word=None):
    """The in-loop dev evaluation of train.py:151-169 as an `eval_score` callable for train(): per dev batch one greedy
    decode through the in-training eval graph (`eval_exe.run(feed={'image': ...}, fetch_list=[caption])`, :163 -- batch
    statistics, and the running statistics ARE updated by it: quirk Q3, the eval program shares the train program's
    variables), the caller's `metric(pred_ids, real_captions) -> float` per batch in place of evaluate.calc_bleu (nltk, out
    of scope: SURVEY.md section 2), the mean over the batches as the score (:165), and the count of distinct filtered
    sentences (:166-167) in the log line.

    dev_batches: callable epoch -> iterable of (images float32 [B,3,S,S], real captions) pairs (the `dev` reader of :152).
    pred_ids handed to `metric` are the float32 id matrix [B, Ti] exactly as fetched (quirk Q2)."""
    from . import decode
    cfg = engine.cfg

    def eval_score(epoch):
        total, n = 0.0, 0
        sentence_said = set()
        for images, real_cap in dev_batches(epoch):
            cp = engine.decode(images).detach().cpu().numpy()                # :163
            total += float(metric(cp, real_cap))                             # :164
            for p in cp.tolist():                                            # :165-166
                sentence_said.add(decode.words2sentence(decode.ids_to_tokens(p, cfg['stop_idx'], cfg['padding_idx'], index_word)))
            n += 1
        score = total / max(1, n)                                            # :167
        if log_path is not None:
            log(log_path, 'Dev set: BLEU 分数: {:.7f} 语句数: {}'.format(score, len(sentence_said)))
        eval_score.sentences = len(sentence_said)
        return score
    return eval_score


def exchange_failure(pg, device, err):
    """Data parallel: every rank learns whether ANY rank failed in this step (a sticky grid-barrier time-out, a NaN loss, a metric
    exception) before the next collective -- a rank that raised alone would leave the others waiting in the next bucket all-reduce
    for ever.  One 4-byte all-reduce per step; every rank then raises (the failing one its own error), which also tears the
    process group down.  pg None: single process, `err` is simply raised."""
    if pg is None:
        if err is not None:
            raise err
        return
    import torch
    import torch.distributed as dist
    flag = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float32, device=device if dist.get_backend(pg) == 'nccl' else 'cpu')
    dist.all_reduce(flag, group=pg)
    if err is not None:
        raise err
    if float(flag.item()) > 0:
        raise RuntimeError('capmi train loop: another rank failed in this step (rank %d stops with it)' % dist.get_rank(pg))


def _rank_world(engine):
    pg = getattr(engine, 'pg', None)
    if pg is None:
        return 0, 1, None
    import torch.distributed as dist
    return dist.get_rank(pg), dist.get_world_size(pg), dist


def train(engine, batches_per_epoch, max_epoch, checkpoint_path, log_path, log_every_n_step=150,
          pretrained_encoder_path=None, checkpoint_backup_every_n_epoch=0, export_params=False,
          trainer=None, eval_score=None, save_best_bleu_checkpoint=True):
    """batches_per_epoch: callable epoch -> iterable of {'image': ..., 'caption': ...} feeds
    (the reader contract of reader.py:45-47,65).  Resumes from `<log_path>/config` like the reference: the epoch is
    written at the START of each epoch (train.py:134), so a crash inside epoch N restarts epoch N from the
    checkpoint written at the end of epoch N-1 (and a crash inside epoch 1 starts from scratch).

    trainer: a dp.OverlappedTrainer around `engine` -- the data-parallel step (train.py:121-124: the SAME
    ParallelExecutor runs the loop, :139); every rank runs this function on its own shard of each batch, rank 0 alone
    writes the checkpoint directory, the resume JSON and the log (one process saves in the reference: :172), every rank
    loads them, and a barrier separates the write from the next read.
    eval_score: callable epoch -> score of the dev evaluation (the reference computes BLEU with nltk, evaluate.py:45-74,
    out of scope here: the CALLER supplies the number).  When it beats `best_bleu` of the resume JSON the persistables are
    also written to `<checkpoint_path>/checkpoint_best_bleu` and the JSON is updated (train.py:85-91, logger.py best_bleu)."""
    rank, world, dist = _rank_world(engine)
    lead = rank == 0

    def barrier():
        if dist is not None and world > 1:
            dist.barrier(group=engine.pg)
    if lead:
        conf = ckpt.load_resume_state(log_path, engine.cfg['encoder_trainable'])       # creates the JSON on first use
    barrier()
    if not lead:
        conf = ckpt.load_resume_state(log_path, engine.cfg['encoder_trainable'])
    if lead:
        load_model(engine, conf, checkpoint_path, log_path, pretrained_encoder_path)
    barrier()
    if not lead:        # reads only: the train_encoder flip (if any) was recorded by rank 0 above
        load_model(engine, dict(conf, train_encoder=bool(engine.cfg['encoder_trainable'])), checkpoint_path, log_path, pretrained_encoder_path)
    step_fn = trainer.train_step if trainer is not None else engine.train_step

    def all_ok(err):
        exchange_failure(engine.pg if dist is not None and world > 1 else None, engine.device, err)

    for epoch in range(conf['epoch'], max_epoch + 1):
        conf['epoch'] = epoch                                   # written at the START of the epoch (train.py:134)
        if lead:
            ckpt.save_resume_state(log_path, conf)
            log(log_path, 'Epoch {}'.format(epoch))
        epoch_loss, step = 0.0, -1
        for step, data in enumerate(batches_per_epoch(epoch)):
            err = None
            try:
                loss, lr = step_fn(data['image'], data['caption'])
                step_loss = loss.detach().cpu().numpy()
                engine.check_sync()
                if np.isnan(step_loss).any():                       # train.py:140-141
                    raise AssertionError('Epoch:{} Step:{} Loss为Nan'.format(epoch, step + 1))
            except Exception as e:                                  # noqa: BLE001 -- re-raised by all_ok on every rank
                err = e
            all_ok(err)
            epoch_loss += float(step_loss[0])
            if lead and (step + 1) % log_every_n_step == 0:
                log(log_path, ' ' * 4 + 'Step {} Mean loss: {:6f} Step loss: {:6f}, lr: {}'.format(
                    step + 1, epoch_loss / (step + 1), float(step_loss[0]), str(np.float32(lr))))
        err, score = None, None
        try:
            score = eval_score(epoch) if eval_score is not None else None             # train.py:151-169 (dev BLEU)
        except Exception as e:                                      # noqa: BLE001
            err = e
        all_ok(err)
        if lead:
            log(log_path, 'Epoch loss: {:7f}'.format(epoch_loss / max(1, step + 1)))
            ckpt.save_persistables(engine, os.path.join(checkpoint_path, 'checkpoint'))   # train.py:172 -> :73
            n = checkpoint_backup_every_n_epoch
            if n and epoch % n == 0:                                                      # :74-76
                ckpt.save_persistables(engine, os.path.join(checkpoint_path, 'checkpoint{}'.format(epoch)))
            if export_params:                                                             # :78-79
                ckpt.save_params(engine, os.path.join(checkpoint_path, 'params'))
            if save_best_bleu_checkpoint and score is not None and score > conf.get('best_bleu', 0):   # :85-88
                conf['best_bleu'] = float(score)
                ckpt.save_resume_state(log_path, conf)                                    # logger.py: the setter saves the JSON
                ckpt.save_persistables(engine, os.path.join(checkpoint_path, 'checkpoint_best_bleu'))
        barrier()                                                                         # nobody runs ahead of the files
    return conf
