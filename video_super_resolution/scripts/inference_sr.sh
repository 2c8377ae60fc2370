#!/bin/bash
# Drop-in for the reference's video_super_resolution/scripts/inference_sr.sh: loops over *.mp4 x prompt lines.
video_folder_path='./input/video'
txt_file_path='./input/text/prompt.txt'
frame_length=32

mapfile -t mp4_files < <(find "$video_folder_path" -type f -name "*.mp4" | sort)
mapfile -t lines < "$txt_file_path"

for i in "${!mp4_files[@]}"; do
    mp4_file="${mp4_files[$i]}"
    line="${lines[$i]:-a good video}"
    python ./video_super_resolution/scripts/inference_sr.py \
        --solver_mode 'fast' --steps 15 \
        --input_path "${mp4_file}" \
        --model_path ./pretrained_weight/light_deg.pt \
        --prompt "${line}" \
        --upscale 4 --max_chunk_len ${frame_length} \
        --file_name "$(basename "$mp4_file")" --save_dir ./results
done
