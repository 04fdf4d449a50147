"""The evaluation loop of the reference, `Evaler.predict_model` (yolov6/core/evaler.py:129-194) with its timing split (`speed_result`,
`eval_speed` :366-372), over the HIP path — the caller-side counterpart SURVEY.md 8(c) asks for:

    uint8 batch from the data loader -> (/255) -> model(imgs)[0] -> non_max_suppression(conf, iou, multi_label=True) -> COCO rows

    loop = EvalLoop(model, conf_thres=0.03, iou_thres=0.65, half=True, ids=coco_ids)
    pred_results = loop.predict_model(dataloader)        # list of {"image_id", "category_id", "bbox", "score"} (evaler.py:411-434)
    loop.eval_speed()                                    # {"pre-process": ms, "inference": ms, "NMS": ms} per image, like the reference logs

Differences from the reference, all inside the same call sequence: `/255` is folded into the first kernel (uint8 images go to the engine as
they are: fold_preprocess=True; False converts like evaler.py:161-163), the NMS result stays on the device and the COCO rows of a batch are
one kernel + one device->host copy (post.py).  Everything else — data loader, COCOeval, plots — is the caller's, unchanged.
"""
import time

import torch

from . import nms as _nms
from . import post as _post


def _time_sync(dev):
    torch.cuda.synchronize(dev)          # yolov6/utils/torch_utils.py:time_sync
    return time.time()


class EvalLoop:
    def __init__(self, model, conf_thres=0.03, iou_thres=0.65, half=True, ids=None, is_coco=True, scale_exact=False, fold_preprocess=True, device=None):
        self.model = model.eval()
        self.conf_thres, self.iou_thres, self.half = conf_thres, iou_thres, half           # tools/eval.py:29-30 defaults
        self.ids, self.is_coco, self.scale_exact = ids, is_coco, scale_exact
        self.fold_preprocess = fold_preprocess
        self.device = device if device is not None else next(model.parameters()).device
        self.speed_result = torch.zeros(4)                                                  # [images, pre-process s, inference s, NMS s] (evaler.py:36)
        if half:
            self.model.half()                                                               # evaler.py:112 (masters stay fp32: Model.half)
        else:
            self.model.float()

    def predict_model(self, dataloader):
        pred_results = []
        dev = self.device
        for imgs, targets, paths, shapes in dataloader:
            # pre-process (evaler.py:160-164)
            t1 = _time_sync(dev)
            imgs = imgs.to(dev, non_blocking=True)
            if not (self.fold_preprocess and imgs.dtype == torch.uint8):
                imgs = imgs.half() if self.half else imgs.float()
                imgs /= 255
            self.speed_result[1] += _time_sync(dev) - t1
            # inference (:167-169); uint8 images: `/255` happens inside the first kernel
            t2 = _time_sync(dev)
            if imgs.dtype == torch.uint8:
                self.model.precision = "fp16" if self.half else "fp32"
            with torch.no_grad():
                outputs, _ = self.model(imgs)
            self.speed_result[2] += _time_sync(dev) - t2
            # post-process (:177-180): the detections stay on the device
            t3 = _time_sync(dev)
            raw = _nms.nms_raw(outputs, self.conf_thres, self.iou_thres, multi_label=True)
            self.speed_result[3] += _time_sync(dev) - t3
            self.speed_result[0] += imgs.shape[0]
            # save result (:187)
            pred_results.extend(_post.convert_to_coco_format(raw, imgs, paths, shapes, self.ids, self.is_coco, self.scale_exact))
        return pred_results

    def eval_speed(self):
        n = max(1.0, self.speed_result[0].item())
        pre, inf, nms_t = (1000.0 * self.speed_result[1:] / n).tolist()
        return {"pre-process": pre, "inference": inf, "NMS": nms_t}
