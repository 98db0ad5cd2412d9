"""
ORACLE (test infrastructure only) -- CPU restatement of the reference's VOC07 11-point average
precision (average_precision.py:30-192).  Pinned: checked against the reference itself imported
in the build container (np.bool / np.int, which numpy >= 1.24 removed, aliased for the import;
tools/make_golden.py) and against tests/golden/g8_average_precision.npz.

Only tests/ may import this module.

Array form: detections (box f32 [n,4] xmin,xmax,ymin,ymax on the 1000 grid, conf f32, cls, sample id)
and ground truth (box f64 [g,4], cls, sample id).  Tie rule for equal confidences: the earlier
detection first (the reference's argsort(-confs) is unstable: tie order is not contractual).
"""
import numpy as np


def iou_plus1(box, arr):
    """ssdutils.jaccard_overlap on float64 (a float32 box is promoted exactly)."""
    box = np.asarray(box, np.float64); a = np.asarray(arr, np.float64)
    areaa = (a[:, 1] - a[:, 0] + 1) * (a[:, 3] - a[:, 2] + 1)
    areab = (box[1] - box[0] + 1) * (box[3] - box[2] + 1)
    w = np.maximum(0, np.minimum(box[1], a[:, 1]) - np.maximum(box[0], a[:, 0]) + 1)
    h = np.maximum(0, np.minimum(box[3], a[:, 3]) - np.maximum(box[2], a[:, 2]) + 1)
    inter = w * h
    return inter / (areab + areaa - inter)


def compute_aps(det_box, det_conf, det_cls, det_sample, gt_box, gt_cls, gt_sample, minoverlap=0.5):
    """{class id: AP} for every class that has ground truth (average_precision.py:84-176)."""
    det_box = np.asarray(det_box, np.float32).reshape(-1, 4); det_conf = np.asarray(det_conf, np.float32)
    det_cls = np.asarray(det_cls); det_sample = np.asarray(det_sample)
    gt_box = np.asarray(gt_box, np.float64).reshape(-1, 4); gt_cls = np.asarray(gt_cls); gt_sample = np.asarray(gt_sample)
    aps = {}
    seen = []
    for c in gt_cls:                     # the reference's dict order: first appearance in the ground truth
        if int(c) not in seen:
            seen.append(int(c))
    for k in seen:
        gsel = np.nonzero(gt_cls == k)[0]
        count = len(gsel)
        matched = np.zeros(len(gt_box), bool)
        dsel = np.nonzero(det_cls == k)[0]
        order = dsel[np.lexsort((dsel, -det_conf[dsel].astype(np.float64)))]      # conf desc, insertion order asc
        tps = np.zeros(len(order)); fps = np.zeros(len(order))
        for i, d in enumerate(order):
            g = gsel[gt_sample[gsel] == det_sample[d]]
            if len(g) == 0:
                fps[i] = 1; continue
            iou = iou_plus1(det_box[d], gt_box[g])
            m = int(np.argmax(iou))
            if iou[m] < minoverlap or matched[g[m]]:
                fps[i] = 1; continue
            tps[i] = 1; matched[g[m]] = True
        fps = np.cumsum(fps); tps = np.cumsum(tps)
        recall = tps / count
        prec = tps / (tps + fps)
        ap = 0
        for r_tilde in np.arange(0, 1.1, 0.1):
            pr = prec[recall >= r_tilde]
            if len(pr) > 0:
                ap += np.amax(pr)
        aps[k] = ap / 11.
    return aps


def aps2map(aps):
    """APs2mAP (average_precision.py:30-42): summed in the dict's order."""
    total = 0.
    for v in aps.values():
        total += v
    return 0 if not aps else total / float(len(aps))
