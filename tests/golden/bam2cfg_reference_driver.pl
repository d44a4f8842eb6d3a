#!/usr/bin/perl
# Runs the parts of the reference's bam2cfg that CAN run in this image, to make golden vectors (tests/golden/make_bam2cfg_perl_vectors.py):
#   * perl/AlnParser.pm, as it is (it has no dependencies): AlnParser::in on every line of a SAM text;
#   * the numeric subs of perl/bam2cfg.pl (ShapiroWilk, ppnd, alnorm, poly_, min, sign, asin; :284-770).  The script itself
#     cannot be loaded -- its `use Statistics::Descriptive` / `use GD::Graph::histogram` lines name modules this image lacks and
#     its records come from a `samtools view` pipe -- so the sub definitions are evaluated from the script's text, read from
#     the reference directory at run time (nothing of it is stored in this repository).
# Its main loop (:48-262) is not run by anything here: it needs those modules.
#
#   perl bam2cfg_reference_driver.pl <reference>/perl aln <alt: 0|1> <file.sam>    -> one line per record: flag qual readlen ori dist readgroup
#   perl bam2cfg_reference_driver.pl <reference>/perl sw <vectors.txt>              -> one line per vector: p-value, then the SWnormality text
use strict;
use warnings;

my ($dir, $mode, @rest) = @ARGV;
die "usage: $0 <reference perl dir> aln|sw ...\n" unless defined $mode;

if ($mode eq 'aln') {
    my ($alt, $sam) = @rest;
    require "$dir/AlnParser.pm";
    my $AP = new AlnParser(platform => undef);
    my %RGplatform;
    open(my $fh, "<", $sam) or die "unable to open $sam\n";
    while (<$fh>) {
        chomp;
        if (/^\@RG/) {                       # as bam2cfg.pl:70-83 fills %RGplatform
            my ($id) = ($_ =~ /ID\:(\S+)/);
            my ($platform) = ($_ =~ /PL\:(\S+)/);
            $RGplatform{$id} = $platform;
            next;
        }
        next if (/^\@/);
        my $t = $AP->in($_, 'sam', \%RGplatform, $alt ? 1 : undef);
        printf "%s\t%s\t%s\t%s\t%s\t%s\n", $t->{flag}, $t->{qual}, $t->{readlen}, $t->{ori}, $t->{dist}, defined($t->{readgroup}) ? $t->{readgroup} : '-';
    }
    close $fh;
}
elsif ($mode eq 'sw') {
    open(my $src, "<", "$dir/bam2cfg.pl") or die "unable to open $dir/bam2cfg.pl\n";
    my @l = <$src>;
    close $src;
    my ($first) = grep { $l[$_] =~ /^sub ShapiroWilk/ } 0 .. $#l;
    my ($last) = grep { $l[$_] =~ /^sub close_samtools/ } 0 .. $#l;
    die "bam2cfg.pl: subs not found\n" unless defined $first && defined $last && $first < $last;
    my $code = join("", @l[$first .. $last - 1]);
    eval "$code; 1" or die "evaluating the subs of bam2cfg.pl: $@";
    open(my $fh, "<", $rest[0]) or die "unable to open $rest[0]\n";
    while (<$fh>) {
        chomp;
        my @data = split;
        @data = sort { $a <=> $b } @data;   # bam2cfg.pl:209
        my $p_value = main::ShapiroWilk(\@data);
        my $text = '';                      # bam2cfg.pl:211-229
        if ($p_value > 0) { $text = sprintf("%.2f", log($p_value) / log(10)); }
        elsif ($p_value == -1) { $text = "data not qualified -1"; }
        elsif ($p_value == -2.1) { $text = "data not qualified -2.1"; }
        elsif ($p_value == -2.2) { $text = "data not qualified -2.2"; }
        elsif ($p_value == -2.3) { $text = "data not qualified -2.3"; }
        elsif ($p_value == 0) { $text = "minus infinity"; }
        printf "%.17g\t%s\n", $p_value, $text;
    }
    close $fh;
}
else { die "unknown mode $mode\n"; }
