#!/usr/bin/env bash
# egs/hificaptain_jp_female/tts1 (Hi-Fi-CAPTAIN ja-JP female, single speaker) on the MI355X path: the reference recipe's interface (same variables, same --option syntax, same directory
# layout: exp/<expname>/{config.yml,stats.h5,tokens.txt,*.pkl} -> exp/<expname>/results/<checkpoint>/<set>/wav/*.wav) with
# stage 4 (network decoding) running on jatts_amd.  Stages -1..3 (download, data preparation, feature extraction,
# statistics, training) and 5 (evaluation) are the reference's and are not rebuilt here: run them there, then point
# --expdir / --checkpoint at the result, or start this script with --stage 4 inside the reference's recipe directory.
#     ./run.sh --stage 4 --stop_stage 4 --tag mytag [--checkpoint exp/.../checkpoint-100000steps.pkl] [--n_gpus 8]

log() {
    local fname=${BASH_SOURCE[1]##*/}
    echo -e "$(date '+%Y-%m-%dT%H:%M:%S') (${fname}:${BASH_LINENO[0]}:${FUNCNAME[1]}) $*"
}

HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REPO_ROOT="$(cd "${HERE}/../../.." && pwd)"
export PYTHONPATH="${REPO_ROOT}${PYTHONPATH:+:${PYTHONPATH}}"
export PYTHONIOENCODING=UTF-8
python=${PYTHON:-python3}

# basic settings (reference egs/hificaptain_jp_female/tts1/run.sh:14-56)
stage=4        # stage to start
stop_stage=4   # stage to stop
verbose=1      # verbosity level (lower is less info)
n_gpus=1       # number of gpus (decoding: one process per GPU, utterances sharded)

conf=conf/fastspeech2.v1.yaml

# text related setting
token_type="phn"
token_column="phonemes"
g2p=julius
cleaner=none

# training related setting
tag=""         # tag for directory to save model
expdir=""      # exp/<expname>; derived from conf / tag like the reference when empty

# decoding related setting
outdir=
checkpoint=""  # checkpoint path to be used for decoding; if not provided, the latest one will be used
precision=fp32        # fp32 = the reference's arithmetic; fp32_bf16x3 = f32 tensors, f32-equivalent emulated MFMA operands (1.35x; fp32_bf16x3_6p: 1.5x);
                      # fp32_split = f32 tensors + split-precision MFMA operands (2.5x); fp16 = fast mode
decode_batch_size=64  # utterances per ragged batch
master_port=29517

# shellcheck disable=SC1091
. "${REPO_ROOT}/egs/common/parse_options.sh" || exit 1
. "${REPO_ROOT}/egs/common/stage4.sh" || exit 1

set -euo pipefail

train_set="train"
dev_set="dev"
test_set="test"
decode_sets="dev_raw_feat ${test_set}"   # reference egs/hificaptain_jp_female/tts1/run.sh: stage 4 decodes the dev features' csv and the test set

if [ -z "${expdir}" ]; then
    if [ -z "${tag}" ]; then
        expname="${train_set}_${token_type}_${cleaner}_$(basename "${conf%.*}")"
    else
        expname="${train_set}_${token_type}_${cleaner}_${tag}"
    fi
    expdir=exp/${expname}
fi

if [ "${stage}" -le 3 ]; then
    log "Stages <= 3 (data preparation, features, statistics, training) are the reference recipe's own: run them there."
fi

if [ "${stage}" -le 4 ] && [ "${stop_stage}" -ge 4 ]; then
    log "Stage 4: Network decoding"
    stage4_decode
fi

if [ "${stage}" -le 5 ] && [ "${stop_stage}" -ge 5 ]; then
    log "Stage 5 (objective evaluation) is the reference recipe's own (evaluate.py on ${expdir}/results/*/${test_set}/wav)."
fi
